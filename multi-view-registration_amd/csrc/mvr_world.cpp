// csrc/mvr_world.cpp -- the multi-GPU host of the global (ring / LUM) registration, behind the C-ABI, no torch:
//
//   * mvr_ring_run_sharded : ONE RANK's part of the outer passes of Registrator::registrationLUM
//     (mvr/src/registrator.cpp:625-664).  The V * Ns source queries of all ring pairs (:640-651, the independent,
//     shardable unit) are dealt to the ranks in contiguous equal ranges (a pair may be split between two ranks: the
//     sums are additive); every rank holds every scan, so no point ever crosses the fabric.  Per pass: pose the scans
//     this rank's ranges touch, run its share of the fused searches + sums into a device table [edges][32] f64,
//     ONE ncclAllReduce(sum) of that table over RCCL/xGMI on the context's stream (3 KB at 12 views: latency-bound),
//     copy it to the host, and solve (per-pair Umeyama, Lu-Milios, pose update) -- redundantly on every rank, the
//     all-reduced bits being identical everywhere, so no broadcast is needed.
//   * one process per GPU (the bench contract): mvr_comm_unique_id on rank 0, the 128 bytes travel by whatever
//     launcher there is (torch.distributed's store, MPI, a file), mvr_ctx_comm_init = ncclCommInitRank.
//   * one process, all GPUs: mvr_world_create = one context per device + ncclCommInitAll;
//     mvr_world_ring_run runs mvr_ring_run_sharded on one host thread per device.
//
// RCCL is bound at run time (dlopen): libmvr_hip.so itself does not link it, so the single-GPU path neither loads nor
// needs it, and a process that already holds an RCCL (PyTorch-ROCm bundles one, built for the HIP runtime it also
// bundles) shares that copy instead of getting a second one.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "mvr_internal.h"

namespace mvr {
namespace {

struct Rccl {
  void *lib = nullptr;
  std::string path, error;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                               // (optional symbols: older libraries)
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl &rccl()
{
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::pair<std::string, int> > tries;
    if (const char *p = std::getenv("MVR_RCCL_LIB")) tries.push_back({p, RTLD_NOW | RTLD_GLOBAL});
    for (const char *n : {"librccl.so", "librccl.so.1"}) tries.push_back({n, RTLD_NOW | RTLD_NOLOAD});      // a copy the process holds already
    for (const char *n : {"librccl.so.1", "librccl.so"}) tries.push_back({n, RTLD_NOW | RTLD_GLOBAL});
    for (auto &t : tries) {
      r.lib = dlopen(t.first.c_str(), t.second);
      if (r.lib) { r.path = t.first; break; }
    }
    if (!r.lib) { r.error = std::string("RCCL not found (librccl.so): ") + (dlerror() ? dlerror() : ""); return; }
#define MVR_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
    MVR_SYM(GetUniqueId, "ncclGetUniqueId"); MVR_SYM(CommInitRank, "ncclCommInitRank"); MVR_SYM(CommInitAll, "ncclCommInitAll");
    MVR_SYM(CommDestroy, "ncclCommDestroy"); MVR_SYM(CommCount, "ncclCommCount"); MVR_SYM(AllReduce, "ncclAllReduce");
    MVR_SYM(GetErrorString, "ncclGetErrorString"); MVR_SYM(CommAbort, "ncclCommAbort"); MVR_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
#undef MVR_SYM
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
      r.error = "RCCL library lacks an expected symbol: " + r.path;
      r.lib = nullptr;
    }
  });
  return r;
}

int rccl_fail(Ctx *c, const char *what, ncclResult_t e)
{
  std::string msg = what;
  if (rccl().GetErrorString) { msg += ": "; msg += rccl().GetErrorString(e); }
  return set_error(c, MVR_E_RCCL, msg.c_str());
}

struct Segment { int edge; size_t q_begin, q_count; };

// the rank's share of the concatenated source queries of all edges: [total rank / world, total (rank + 1) / world)
std::vector<Segment> split_queries(const std::vector<size_t> &sizes, int world, int rank)
{
  unsigned long long total = 0;
  for (size_t n : sizes) total += n;
  const unsigned long long lo = total * (unsigned long long)rank / (unsigned long long)world;
  const unsigned long long hi = total * (unsigned long long)(rank + 1) / (unsigned long long)world;
  std::vector<Segment> out;
  unsigned long long base = 0;
  for (size_t e = 0; e < sizes.size(); ++e) {
    const unsigned long long a = std::max(lo, base), b = std::min(hi, base + sizes[e]);
    if (b > a) out.push_back(Segment{(int)e, (size_t)(a - base), (size_t)(b - a)});
    base += sizes[e];
  }
  return out;
}

}  // namespace

// ---- what the rest of the library sees of a communicator (declared in mvr_internal.h) ----------------------------------
// Failure model.  A collective needs every rank; RCCL neither notices a peer that died nor one that left its loop early.
// (1) A rank whose LOCAL work fails does not leave: it joins the pass's collectives with neutral contributions and a
//     raised failure count in the reduced table, so all ranks end that pass together, each with an error.
// (2) A rank that waits for a pass containing a collective waits at most wait_timeout_ms, polling RCCL's asynchronous error
//     state meanwhile; on either it aborts the communicator (ncclCommAbort: the queued collective kernels exit), drains
//     its stream and returns MVR_E_RCCL.  The context stays usable for single-GPU work; the communicator is gone.
static bool trace_comm() { static const bool on = std::getenv("MVR_TRACE_COMM") != nullptr; return on; }
#define MVR_TRACE(...) do { if (trace_comm()) { std::fprintf(stderr, "[mvr comm] " __VA_ARGS__); std::fprintf(stderr, "\n"); std::fflush(stderr); } } while (0)

int comm_abort(Ctx *c, const char *why)
{
  MVR_TRACE("abort: %s (comm %p, stall word %p)", why, c->comm, (void *)c->h_stall);
  Rccl &r = rccl();
  if (c->h_stall) __atomic_store_n(c->h_stall, 1u, __ATOMIC_RELEASE);            // (an injected stall must not outlive the abort)
  if (c->comm) {
    if (r.lib && r.CommAbort) {
      // ncclCommAbort itself must not become the hang it is there to end: it runs on a helper thread that is given five
      // seconds; a call that has not come back by then is left behind (the communicator is unusable either way)
      struct Box { std::mutex m; std::condition_variable cv; bool done = false; };
      auto box = std::make_shared<Box>();
      const ncclComm_t comm = reinterpret_cast<ncclComm_t>(c->comm);
      const int device = c->device;
      auto fn = r.CommAbort;
      std::thread([box, comm, device, fn] {
        (void)hipSetDevice(device);
        (void)fn(comm);
        { std::lock_guard<std::mutex> g(box->m); box->done = true; }
        box->cv.notify_all();
      }).detach();
      std::unique_lock<std::mutex> lk(box->m);
      const bool back = box->cv.wait_for(lk, std::chrono::seconds(5), [&] { return box->done; });
      MVR_TRACE("ncclCommAbort %s", back ? "returned" : "did NOT return within 5 s (left behind)");
      // (a call that did not come back leaves its collective on the stream: nothing may wait for that stream without a bound
      // any more -- drain_bounded -- and the process should end rather than go on with this device)
      if (!back) c->stream_stuck = true;
    }
    if (c->comm_lender) *c->comm_lender = nullptr;                                // a world's communicator: the world must not destroy it again
    c->comm = nullptr; c->comm_owned = false; c->comm_lender = nullptr;
    c->comm_broken = true;
  }
  return set_error(c, MVR_E_RCCL, why);
}

// waits for the context's stream, but not for ever once an abort has failed to free it: true = drained
bool drain_bounded(Ctx *c, int ms)
{
  if (!c->stream_stuck) return hipStreamSynchronize(c->stream) == hipSuccess;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipStreamQuery(c->stream);
    if (e == hipSuccess) return true;
    if (e != hipErrorNotReady) return false;
    if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > (double)ms) {
      MVR_TRACE("the stream did not drain within %d ms behind an abort that never returned: left as it is", ms);
      return false;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

int comm_poll(Ctx *c, bool abort_now)
{
  if (!c->comm) return MVR_OK;
  Rccl &r = rccl();
  if (!r.lib || !r.CommGetAsyncError) return MVR_OK;
  ncclResult_t st = ncclSuccess;
  if (r.CommGetAsyncError(reinterpret_cast<ncclComm_t>(c->comm), &st) != ncclSuccess || (st != ncclSuccess && st != ncclInProgress)) {
    std::string msg = "RCCL reports an asynchronous error";
    if (r.GetErrorString) { msg += ": "; msg += r.GetErrorString(st); }
    // (a caller with a chain queued behind a gate opens the gate FIRST: ncclCommAbort waits for the stream, the stream for the gate)
    return abort_now ? comm_abort(c, msg.c_str()) : set_error(c, MVR_E_RCCL, msg.c_str());
  }
  return MVR_OK;
}

int stream_wait(Ctx *c)
{
  if (!c->comm && !c->h_stall) { MVR_HIP_TRY(c, hipStreamSynchronize(c->stream)); return MVR_OK; }
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  unsigned n = 0;
  MVR_TRACE("stream_wait: bounded, %d ms", c->wait_timeout_ms);
  for (;;) {
    const hipError_t e = hipStreamQuery(c->stream);
    if (e == hipSuccess) return MVR_OK;
    if (e != hipErrorNotReady) return set_error(c, MVR_E_HIP, "hipStreamQuery", e);
    if ((++n & 0x3FFu) == 0u) {
      if (int rc = comm_poll(c)) { (void)drain_bounded(c, 5000); return rc; }
      if (std::chrono::duration<double, std::milli>(clk::now() - t0).count() > (double)c->wait_timeout_ms) {
        const int rc = comm_abort(c, "a pass with a collective did not finish in time: a peer failed or never arrived");
        MVR_TRACE("draining the stream");
        (void)drain_bounded(c, 5000);                                             // the aborted collective and everything behind it drain now
        MVR_TRACE("drained");
        return rc;
      }
    }
    __builtin_ia32_pause();
  }
}

int comm_allreduce(Ctx *c, void *dev_buf, size_t count, int kind)
{
  const bool stall = c->inject_stall_at >= 0 && c->dist_pass - 1 == c->inject_stall_at && kind == kReduceSumF64;      // (the LAST collective of a pass / iteration)
  auto stall_stream = [&]() -> int {
    // test hook: the pass this collective belongs to never finishes on this rank -- what a peer that died inside the
    // collective looks like from here (the word is released by comm_abort).  The wait is queued BEHIND the collective:
    // RCCL's enqueue may itself synchronise with the stream and must not meet a stream that cannot move.
    if (!c->h_stall) {
      MVR_HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_stall), 64, hipHostMallocMapped));
      MVR_HIP_TRY(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_stall), c->h_stall, 0));
    }
    *c->h_stall = 0u;
    c->inject_stall_at = -1;
    MVR_TRACE("injecting a stall behind the collective");
    MVR_HIP_TRY(c, hipStreamWaitValue32(c->stream, c->d_stall, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu));
    MVR_TRACE("stall queued");
    return MVR_OK;
  };
  if (!c->comm || count == 0) return stall ? stall_stream() : MVR_OK;
  Rccl &r = rccl();
  if (!r.lib) return set_error(c, MVR_E_RCCL, r.error.c_str());
  const ncclResult_t e = kind == kReduceMinI64
      ? r.AllReduce(dev_buf, dev_buf, count, ncclInt64, ncclMin, reinterpret_cast<ncclComm_t>(c->comm), c->stream)
      : r.AllReduce(dev_buf, dev_buf, count, ncclDouble, ncclSum, reinterpret_cast<ncclComm_t>(c->comm), c->stream);
  if (e != ncclSuccess) {
    std::string msg = "ncclAllReduce";
    if (r.GetErrorString) { msg += ": "; msg += r.GetErrorString(e); }
    return comm_abort(c, msg.c_str());
  }
  return stall ? stall_stream() : MVR_OK;
}

}  // namespace mvr

using namespace mvr;

#define API __attribute__((visibility("default")))
#define CTX(p) reinterpret_cast<Ctx *>(p)

struct mvr_world {
  std::vector<mvr_ctx *> ctx;
  std::vector<ncclComm_t> comm;        // ncclCommInitAll's communicators (owned here, lent to the contexts)
  std::string last_error;
};

extern "C" {

API int mvr_comm_unique_id(char id[MVR_UNIQUE_ID_BYTES])
{
  static_assert(MVR_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  if (!id) return MVR_E_ARG;
  Rccl &r = rccl();
  if (!r.lib) return MVR_E_RCCL;
  ncclUniqueId u;
  if (r.GetUniqueId(&u) != ncclSuccess) return MVR_E_RCCL;
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return MVR_OK;
}

API const char *mvr_rccl_library(void)
{
  Rccl &r = rccl();
  return r.lib ? r.path.c_str() : r.error.c_str();
}

API int mvr_ctx_comm_init(mvr_ctx *ctx, const char id[MVR_UNIQUE_ID_BYTES], int rank, int world)
{
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  if (c->comm) return set_error(c, MVR_E_ARG, "this context already has a communicator");
  Rccl &r = rccl();
  if (!r.lib) return set_error(c, MVR_E_RCCL, r.error.c_str());
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  const ncclResult_t e = r.CommInitRank(&comm, world, u, rank);
  if (e != ncclSuccess) return rccl_fail(c, "ncclCommInitRank", e);
  c->comm = comm; c->comm_owned = true; c->comm_rank = rank; c->comm_world = world; c->comm_lender = nullptr; c->comm_broken = false;
  return MVR_OK;
}

API int mvr_ctx_comm_destroy(mvr_ctx *ctx)
{
  if (!ctx) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  if (c->comm && c->comm_owned && rccl().lib) {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)rccl().CommDestroy(reinterpret_cast<ncclComm_t>(c->comm));
  }
  c->comm = nullptr; c->comm_owned = false; c->comm_rank = 0; c->comm_world = 1; c->comm_lender = nullptr; c->comm_broken = false;
  return MVR_OK;
}

API int mvr_ctx_comm_info(mvr_ctx *ctx, int *rank, int *world, int *rccl_ranks)
{
  if (!ctx) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  if (rank) *rank = c->comm_rank;
  if (world) *world = c->comm_world;
  if (rccl_ranks) {
    *rccl_ranks = 0;          // what RCCL itself says (0: no communicator)
    if (c->comm && rccl().CommCount) (void)rccl().CommCount(reinterpret_cast<ncclComm_t>(c->comm), rccl_ranks);
  }
  return MVR_OK;
}

API int mvr_ring_segments(int ne, const size_t *edge_queries, int world, int rank, int *seg_edge, size_t *seg_begin, size_t *seg_count, int *n_seg)
{
  if (ne < 0 || (ne && !edge_queries) || world < 1 || rank < 0 || rank >= world || !n_seg) return MVR_E_ARG;
  const std::vector<Segment> s = split_queries(std::vector<size_t>(edge_queries, edge_queries + ne), world, rank);
  *n_seg = (int)s.size();
  for (size_t k = 0; k < s.size(); ++k) {
    if (seg_edge) seg_edge[k] = s[k].edge;
    if (seg_begin) seg_begin[k] = s[k].q_begin;
    if (seg_count) seg_count[k] = s[k].q_count;
  }
  return MVR_OK;
}

namespace {
// what one rank does in every pass: which scans to pose, which (pair, query range) to search
struct RankPlan {
  std::vector<Segment> segs;
  std::vector<int> ss, ts, vdst, vsrc, vview;
  std::vector<size_t> qb, qn;
};

int make_rank_plan(mvr_ctx *ctx, int rank, int world, int n_views, const int *posed_slots, const int *raw_slots, int ne, const int *edge_src,
                   const int *edge_tgt, RankPlan *p)
{
  std::vector<size_t> sizes((size_t)ne);
  for (int e = 0; e < ne; ++e) {             // edge e's queries are the points of its SOURCE scan
    size_t n = 0;
    if (int rc = mvr_cloud_size(ctx, raw_slots[edge_src[e]], &n)) return rc;
    sizes[(size_t)e] = n;
  }
  p->segs = split_queries(sizes, world, rank);
  std::vector<char> need((size_t)n_views, 0);
  for (const Segment &s : p->segs) {
    p->ss.push_back(posed_slots[edge_src[s.edge]]); p->ts.push_back(posed_slots[edge_tgt[s.edge]]);
    p->qb.push_back(s.q_begin); p->qn.push_back(s.q_count);
    need[(size_t)edge_src[s.edge]] = need[(size_t)edge_tgt[s.edge]] = 1;
  }
  for (int v = 0; v < n_views; ++v) if (need[(size_t)v]) { p->vdst.push_back(posed_slots[v]); p->vsrc.push_back(raw_slots[v]); p->vview.push_back(v); }
  return MVR_OK;
}

// pose the rank's scans and put its rows of the edge table (zero elsewhere) into c->dist_table, asynchronously
int enqueue_rank_rows(mvr_ctx *ctx, const RankPlan &p, int ne, double max_dist, int reciprocal, int fma, const double origin[3], const double *poses)
{
  Ctx *c = CTX(ctx);
  std::vector<double> T(p.vdst.size() * 16);
  for (size_t k = 0; k < p.vview.size(); ++k) std::memcpy(&T[k * 16], poses + 16 * (size_t)p.vview[k], 16 * sizeof(double));
  if (!p.vdst.empty()) { if (int rc = mvr_cloud_transform_batch(ctx, (int)p.vdst.size(), p.vdst.data(), p.vsrc.data(), T.data())) return rc; }
  MVR_HIP_TRY(c, hipMemsetAsync(c->dist_table, 0, (size_t)std::max(ne, 1) * 32 * sizeof(double), c->stream));       // rows of other ranks' edges: zero
  if (!p.segs.empty()) {
    // a rank's ranges cover consecutive edges: rows e0 .. of the table, one batched call (one launch per stage)
    if (int rc = mvr_pair_moments2_batch(ctx, (int)p.segs.size(), p.ss.data(), p.ts.data(), max_dist, reciprocal, fma, p.qb.data(), p.qn.data(), origin,
                                         nullptr, c->dist_table + (size_t)p.segs[0].edge * 32)) return rc;
  }
  return MVR_OK;
}

int ensure_tables(Ctx *c, int ne)
{
  const size_t table_n = (size_t)std::max(ne, 1) * 32;
  if (int rc = ensure(c, c->dist_table, c->dist_table_cap, table_n)) return rc;
  if (c->h_table_cap < table_n) {
    if (c->h_table) { MVR_HIP_TRY(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_table); }      // (only an OLD table can still be written to)
    c->h_table = nullptr; c->h_table_cap = 0;
    MVR_HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_table), std::max(table_n, (size_t)512) * sizeof(double), hipHostMallocMapped));
    c->h_table_cap = std::max(table_n, (size_t)512);
  }
  return MVR_OK;
}

bool ring_args_ok(int n_views, const int *posed_slots, const int *raw_slots, int ne, const int *edge_src, const int *edge_tgt)
{
  if (n_views < 2 || ne < 0 || !posed_slots || !raw_slots || (ne && (!edge_src || !edge_tgt))) return false;
  for (int e = 0; e < ne; ++e) if (edge_src[e] < 0 || edge_src[e] >= n_views || edge_tgt[e] < 0 || edge_tgt[e] >= n_views) return false;
  return true;
}
}  // namespace

API int mvr_ctx_project(mvr_ctx *ctx, int world, int rank, const double *peer_rows, int ne)
{
  if (!ctx) return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  if (world <= 1) { c->project_world = 0; c->project_rank = 0; c->proj_rows_n = 0; return MVR_OK; }
  if (rank < 0 || rank >= world || ne < 0 || (ne && !peer_rows)) return MVR_E_ARG;
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = (size_t)ne * 32;
  if (int rc = ensure(c, c->proj_rows, c->proj_rows_cap, std::max(n, (size_t)32))) return rc;
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (n) MVR_HIP_TRY(c, hipMemcpy(c->proj_rows, peer_rows, n * sizeof(double), hipMemcpyHostToDevice));
  c->proj_rows_n = n; c->project_world = world; c->project_rank = rank;
  return MVR_OK;
}

API int mvr_ring_rows_sharded(mvr_ctx *ctx, int rank, int world, int n_views, const int *posed_slots, const int *raw_slots, int ne,
                              const int *edge_src, const int *edge_tgt, double max_dist, int reciprocal, int fma, const double origin[3],
                              const double *poses, double *rows)
{
  if (!ctx || world < 1 || rank < 0 || rank >= world || !ring_args_ok(n_views, posed_slots, raw_slots, ne, edge_src, edge_tgt) || !origin || !poses || !rows)
    return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  RankPlan plan;
  if (int rc = make_rank_plan(ctx, rank, world, n_views, posed_slots, raw_slots, ne, edge_src, edge_tgt, &plan)) return rc;
  if (int rc = ensure_tables(c, ne)) return rc;
  if (int rc = enqueue_rank_rows(ctx, plan, ne, max_dist, reciprocal, fma, origin, poses)) return rc;
  if (ne) MVR_HIP_TRY(c, hipMemcpyAsync(c->h_table, c->dist_table, (size_t)ne * 32 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MVR_HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (ne) std::memcpy(rows, c->h_table, (size_t)ne * 32 * sizeof(double));
  return MVR_OK;
}

namespace {
// one rank's pass of the sharded ring run, in the two halves ring_passes drives
struct RankRun {
  mvr_ctx *ctx; RankPlan plan; int n_views, ne; const int *edge_src, *edge_tgt; double max_dist; int reciprocal, fma; const double *origin;
  int lum_iterations; double *poses, *lum_pose; float *pair_T; double *pair_n, *pair_mse; int *lum_iters; double *rows;
  std::vector<double> pn, pm;
  int rank = 0, world = 1;
  int passes_left = 1;      // (pair_T only for the last pass of the run: RingRun, mvr_ctx.hip)
  static int enqueue(void *p)
  {
    RankRun &r = *static_cast<RankRun *>(p);
    Ctx *c = CTX(r.ctx);
    const size_t table_n = (size_t)(r.ne + 1) * 32;                 // the edge rows + one status row ([0] = ranks whose local work failed)
    int local = MVR_OK;
    if (c->inject_fail_at >= 0 && c->dist_pass == c->inject_fail_at) {
      if (c->no_sync) return set_error(c, MVR_E_HIP, "not in steady state: injected failure");      // (it strikes when the pass is enqueued the ordinary way)
      local = set_error(c, MVR_E_HIP, "injected failure of this rank's local work");
    }
    ++c->dist_pass;
    if (local == MVR_OK) local = enqueue_rank_rows(r.ctx, r.plan, r.ne, r.max_dist, r.reciprocal, r.fma, r.origin, r.poses);
    // a chain that is being queued ahead of its poses and turns out not to be in steady state is simply enqueued again the
    // ordinary way (ring_passes): nothing of it may reach the collective.  Any OTHER failure of the local work is reported
    // THROUGH the collective, so that the peers end this pass with an error too instead of waiting in it for ever.
    if (local != MVR_OK && c->no_sync) return local;
    // the status row: [0] = ranks that failed, [1 + rank] = this rank's (negated) status -- so that after the reduction a rank
    // knows whether the failure was its own without any host-side memory of the pass (passes are enqueued ahead of their solves)
    MVR_HIP_TRY(c, hipMemsetAsync(c->dist_table + (size_t)r.ne * 32, 0, 32 * sizeof(double), c->stream));
    if (local != MVR_OK) {
      MVR_HIP_TRY(c, hipMemsetAsync(c->dist_table, 0, (size_t)r.ne * 32 * sizeof(double), c->stream));
      double *flag = c->h_moments + 48;      // (pinned; only ever written in this branch, which is never taken while a chain is queued ahead)
      flag[0] = 1.0; flag[1] = (double)(-local);
      MVR_HIP_TRY(c, hipMemcpyAsync(c->dist_table + (size_t)r.ne * 32, flag, sizeof(double), hipMemcpyHostToDevice, c->stream));
      MVR_HIP_TRY(c, hipMemcpyAsync(c->dist_table + (size_t)r.ne * 32 + 1 + (size_t)(r.rank % 31), flag + 1, sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if (int rc = comm_allreduce(c, c->dist_table, table_n, kReduceSumF64)) return rc;
    // (projection of one rank's share, mvr_ctx_project: what the absent peers would have added)
    if (c->project_world > 1 && c->proj_rows_n == (size_t)r.ne * 32) { if (int rc = launch_add_f64(c, c->dist_table, c->proj_rows, c->proj_rows_n)) return rc; }
    MVR_HIP_TRY(c, hipMemcpyAsync(c->h_table, c->dist_table, table_n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return MVR_OK;
  }
  static int solve(void *p)
  {
    RankRun &r = *static_cast<RankRun *>(p);
    Ctx *c = CTX(r.ctx);
    if (c->h_table[(size_t)r.ne * 32] > 0.0) {
      const double mine = c->h_table[(size_t)r.ne * 32 + 1 + (size_t)(r.rank % 31)];
      if (mine > 0.0 && r.world <= 31) return set_error(c, -(int)mine, "this rank's local work failed in this pass (reported to its peers through the pass's all-reduce)");
      return set_error(c, MVR_E_RCCL, "a peer's local work failed in this pass");
    }
    if (r.rows && r.ne) std::memcpy(r.rows, c->h_table, (size_t)r.ne * 32 * sizeof(double));
    const int rc = mvr_ring_host_step(r.n_views, r.ne, r.edge_src, r.edge_tgt, c->h_table, r.origin, r.lum_iterations, r.poses, r.lum_pose, --r.passes_left <= 0 ? r.pair_T : nullptr,
                                      r.pair_n ? r.pair_n : r.pn.data(), r.pair_mse ? r.pair_mse : r.pm.data(), r.lum_iters);
    return rc != MVR_OK ? set_error(c, rc, "LUM solve") : MVR_OK;
  }
};
}  // namespace

API int mvr_ring_run_sharded(mvr_ctx *ctx, int n_steps, int n_views, const int *posed_slots, const int *raw_slots, int ne,
                             const int *edge_src, const int *edge_tgt, double max_dist, int reciprocal, int fma, const double origin[3],
                             int lum_iterations, double *poses, double *lum_pose, float *pair_T, double *pair_n, double *pair_mse,
                             int *lum_iters, double *rows, double *timing_ms)
{
  if (!ctx || n_steps < 0 || !ring_args_ok(n_views, posed_slots, raw_slots, ne, edge_src, edge_tgt) || !origin || !poses || !lum_pose)
    return MVR_E_ARG;
  Ctx *c = CTX(ctx);
  MVR_HIP_TRY(c, hipSetDevice(c->device));
  if (c->comm_broken) return set_error(c, MVR_E_RCCL, "the communicator of this context was aborted");
  int world = c->comm ? c->comm_world : 1, rank = c->comm ? c->comm_rank : 0;
  if (c->project_world > 1) {      // a projection: this context plays ONE rank of a larger world (the communicator, if any, stays what it is)
    if (world != 1) return set_error(c, MVR_E_ARG, "mvr_ctx_project needs a context that is a world of one");
    world = c->project_world; rank = c->project_rank;
  }
  Rccl &r = rccl();
  if (c->comm && !r.lib) return set_error(c, MVR_E_RCCL, r.error.c_str());
  // everything that can fail for local reasons and is known up front happens BEFORE the first collective: the plan,
  // the tables (the passes themselves report later failures through the collective, see RankRun::enqueue)
  RankRun run{ctx, RankPlan(), n_views, ne, edge_src, edge_tgt, max_dist, reciprocal, fma, origin, lum_iterations, poses, lum_pose, pair_T,
              pair_n, pair_mse, lum_iters, rows, std::vector<double>((size_t)ne), std::vector<double>((size_t)ne), rank, world};
  run.passes_left = n_steps;
  if (int rc = make_rank_plan(ctx, rank, world, n_views, posed_slots, raw_slots, ne, edge_src, edge_tgt, &run.plan)) return rc;
  if (int rc = ensure_tables(c, ne + 1)) return rc;
  PassLoop L;
  L.n_views = n_views; L.posed_slots = posed_slots; L.raw_slots = raw_slots; L.poses = poses;
  L.enqueue = &RankRun::enqueue; L.solve = &RankRun::solve; L.self = &run;
  L.sig = pass_loop_sig(c, n_views, posed_slots, raw_slots, ne, edge_src, edge_tgt, max_dist, reciprocal, fma, 1 + rank + 1000 * world);
  L.reach = max_dist;
  if (int rc = ring_passes(c, n_steps, L, timing_ms)) return rc;
  // Every rank solved the SAME all-reduced table and so holds the same poses -- if RCCL handed every rank the same bits (one
  // reduction order per element, whatever the algorithm).  That assumption is checked, once per call, not taken on trust:
  // a hash of the poses and its complement are MIN-reduced; all ranks agree exactly when min(h) == ~min(~h).
  if (c->comm && n_steps > 0) {
    unsigned long long h = 1469598103934665603ull;
    const unsigned char *b = reinterpret_cast<const unsigned char *>(poses);
    for (size_t i = 0; i < (size_t)n_views * 16 * sizeof(double); ++i) { h ^= b[i]; h *= 1099511628211ull; }
    h >>= 1;                                              // (non-negative as a signed 64-bit value: ncclInt64 / ncclMin)
    if (int rc = ensure(c, c->seq_keys, c->seq_keys_cap, 2)) return rc;
    long long *hk = reinterpret_cast<long long *>(c->h_moments + 50);
    hk[0] = (long long)h; hk[1] = (long long)(0x7FFFFFFFFFFFFFFFull - h);
    MVR_HIP_TRY(c, hipMemcpyAsync(c->seq_keys, hk, 2 * sizeof(long long), hipMemcpyHostToDevice, c->stream));
    if (int rc = comm_allreduce(c, c->seq_keys, 2, kReduceMinI64)) return rc;
    MVR_HIP_TRY(c, hipMemcpyAsync(hk, c->seq_keys, 2 * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    if (int rc = stream_wait(c)) return rc;
    if ((unsigned long long)hk[0] != 0x7FFFFFFFFFFFFFFFull - (unsigned long long)hk[1])
      return set_error(c, MVR_E_RCCL, "the ranks disagree on the poses: the all-reduce did not hand every rank the same bits");
  }
  return MVR_OK;
}

// ------------------------------------------------------------------ one process, all GPUs

API int mvr_world_create(mvr_world **out, int n_dev, const int *device_ids)
{
  if (!out || n_dev < 1) return MVR_E_ARG;
  *out = nullptr;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return MVR_E_HIP;
  std::vector<int> devs((size_t)n_dev);
  for (int k = 0; k < n_dev; ++k) {
    devs[(size_t)k] = device_ids ? device_ids[k] : k;
    if (devs[(size_t)k] < 0 || devs[(size_t)k] >= have) return MVR_E_ARG;
    for (int j = 0; j < k; ++j) if (devs[(size_t)j] == devs[(size_t)k]) return MVR_E_ARG;      // one rank per GPU
  }
  Rccl &r = rccl();
  if (!r.lib) return MVR_E_RCCL;
  mvr_world *w = new (std::nothrow) mvr_world();
  if (!w) return MVR_E_NOMEM;
  w->ctx.assign((size_t)n_dev, nullptr);
  for (int k = 0; k < n_dev; ++k)
    if (int rc = mvr_ctx_create(&w->ctx[(size_t)k], devs[(size_t)k])) { mvr_world_destroy(w); return rc; }
  w->comm.assign((size_t)n_dev, nullptr);
  if (r.CommInitAll(w->comm.data(), n_dev, devs.data()) != ncclSuccess) { w->comm.clear(); mvr_world_destroy(w); return MVR_E_RCCL; }
  for (int k = 0; k < n_dev; ++k) {
    Ctx *c = CTX(w->ctx[(size_t)k]);
    c->comm = w->comm[(size_t)k]; c->comm_owned = false; c->comm_rank = k; c->comm_world = n_dev;
    c->comm_lender = reinterpret_cast<void **>(&w->comm[(size_t)k]);
  }
  *out = w;
  return MVR_OK;
}

API int mvr_world_destroy(mvr_world *w)
{
  if (!w) return MVR_E_ARG;
  for (mvr_ctx *c : w->ctx) if (c) { (void)mvr_ctx_sync(c); CTX(c)->comm = nullptr; CTX(c)->comm_lender = nullptr; }
  for (ncclComm_t cm : w->comm) if (cm && rccl().lib) (void)rccl().CommDestroy(cm);
  for (mvr_ctx *c : w->ctx) if (c) (void)mvr_ctx_destroy(c);
  delete w;
  return MVR_OK;
}

API int mvr_world_size(const mvr_world *w) { return w ? (int)w->ctx.size() : 0; }
API mvr_ctx *mvr_world_ctx(mvr_world *w, int rank) { return (w && rank >= 0 && rank < (int)w->ctx.size()) ? w->ctx[(size_t)rank] : nullptr; }
API const char *mvr_world_last_error(const mvr_world *w) { return w ? w->last_error.c_str() : ""; }

// every rank holds every scan: raw scan v goes to slot raw_slots[v] of every context
API int mvr_world_upload(mvr_world *w, int slot, const float *xyz, size_t n, size_t stride_bytes)
{
  if (!w) return MVR_E_ARG;
  for (mvr_ctx *c : w->ctx)
    if (int rc = mvr_cloud_upload(c, slot, xyz, n, stride_bytes)) { w->last_error = mvr_last_error(c); return rc; }
  return MVR_OK;
}

API int mvr_world_ring_run(mvr_world *w, int n_steps, int n_views, const int *posed_slots, const int *raw_slots, int ne, const int *edge_src,
                           const int *edge_tgt, double max_dist, int reciprocal, int fma, const double origin[3], int lum_iterations,
                           double *poses, double *lum_pose, float *pair_T, double *pair_n, double *pair_mse, int *lum_iters, double *rows,
                           double *timing_ms)
{
  if (!w || w->ctx.empty() || n_views < 2 || !poses || !lum_pose) return MVR_E_ARG;
  const size_t G = w->ctx.size();
  // one host thread per device; rank 0 writes the caller's outputs, the others their own copies of the (identical) poses
  std::vector<int> status(G, MVR_OK);
  std::vector<std::vector<double> > P(G, std::vector<double>(poses, poses + (size_t)n_views * 16)), L(G, std::vector<double>((size_t)n_views * 6));
  auto body = [&](size_t g) {
    const bool lead = g == 0;
    status[g] = mvr_ring_run_sharded(w->ctx[g], n_steps, n_views, posed_slots, raw_slots, ne, edge_src, edge_tgt, max_dist, reciprocal, fma, origin,
                                     lum_iterations, lead ? poses : P[g].data(), lead ? lum_pose : L[g].data(), lead ? pair_T : nullptr,
                                     lead ? pair_n : nullptr, lead ? pair_mse : nullptr, lead ? lum_iters : nullptr, lead ? rows : nullptr,
                                     lead ? timing_ms : nullptr);
  };
  std::vector<std::thread> th;
  for (size_t g = 1; g < G; ++g) th.emplace_back(body, g);
  body(0);
  for (std::thread &t : th) t.join();
  for (size_t g = 0; g < G; ++g)
    if (status[g] != MVR_OK) { w->last_error = "rank " + std::to_string(g) + ": " + mvr_last_error(w->ctx[g]); return status[g]; }
  for (size_t g = 1; g < G; ++g)      // the ranks solved the same table: bit-identical poses, or something is badly wrong
    if (std::memcmp(P[g].data(), poses, (size_t)n_views * 16 * sizeof(double)) != 0) { w->last_error = "ranks disagree on the poses"; return MVR_E_RCCL; }
  return MVR_OK;
}

}  // extern "C"
