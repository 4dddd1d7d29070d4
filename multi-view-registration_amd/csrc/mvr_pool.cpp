// csrc/mvr_pool.cpp -- a process-wide cache of the library's device and pinned-host blocks.
//
// A context makes ~50 hipMalloc and ~8 hipHostMalloc calls in the first pass of a registration (work buffers sized by the scans,
// sorted copies, boxes, tables, the pinned tables of the pass loop): 10-15 us each for device memory, ~100 us for pinned memory --
// 0.6-0.7 ms of a first pass that takes 2.6 ms, and hipFree is a device-wide synchronisation on top.  A process that
// registers one object after the other (or a test session: hundreds of contexts) asks for the same sizes again and again.  So every
// hipMalloc / hipFree / hipHostMalloc / hipHostFree of the library goes through here (mvr_internal.h redirects the names): a freed
// block is kept, by device and size class, and handed to the next request of its class; nothing changes for the caller -- a block
// is at least as large as asked for, its contents are undefined as they always were.
//   * a block enters the cache only after the device has drained (hipDeviceSynchronize: what hipFree did implicitly), so whoever
//     gets it next may use it on any stream;
//   * size classes: multiples of 1/8 of the enclosing power of two (a request gets at most 12.5 % more than it asked for); a cached
//     block serves requests of ITS class only (a larger block handed to a smaller request would be missing when its own request
//     comes: identical workloads then find every block they freed);
//   * the cache holds at most MVR_POOL_CAP_MB (default 16384) per process: beyond that a freed block goes back to the runtime; when
//     the runtime is out of memory the cache is emptied and the request repeated; mvr_pool_trim() empties it on demand;
//   * MVR_POOL=0: every call goes straight to the runtime (tools/leak_probe.py, debugging).
// The cache is never destroyed (no static destructor: the HIP runtime may be gone by then); the process's exit frees the memory.
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "mvr_internal.h"

#undef hipMalloc
#undef hipFree
#undef hipHostMalloc
#undef hipHostFree

namespace mvr {
namespace {
struct Key {
  int device; unsigned flags; size_t bytes;      // flags: 0xFFFFFFFF = device memory, else the hipHostMalloc flags of a pinned block
  bool operator<(const Key &o) const { return device != o.device ? device < o.device : flags != o.flags ? flags < o.flags : bytes < o.bytes; }
};
struct Pool {
  std::mutex mu;
  std::multimap<Key, void *> idle;
  std::unordered_map<void *, Key> live;
  size_t cached = 0, cap = 0;
  bool on = true, debug = false;
  unsigned long long hits = 0, misses = 0;
  Pool()
  {
    if (const char *e = std::getenv("MVR_POOL")) on = std::atoi(e) != 0;
    debug = std::getenv("MVR_POOL_DEBUG") != nullptr;      // every request the cache could not serve, on stderr
    size_t mb = 16384;
    if (const char *e = std::getenv("MVR_POOL_CAP_MB")) mb = (size_t)std::max(0, std::atoi(e));
    cap = mb << 20;
  }
};
Pool &pool() { static Pool *p = new Pool(); return *p; }
constexpr unsigned kDeviceMem = 0xFFFFFFFFu;

size_t size_class(size_t n)
{
  if (n <= 4096) return 4096;
  size_t p2 = 4096;
  while (p2 < n) p2 <<= 1;
  const size_t step = p2 >> 3;
  return (n + step - 1) / step * step;
}
hipError_t raw_alloc(void **p, size_t bytes, unsigned flags)
{
  return flags == kDeviceMem ? hipMalloc(p, bytes) : hipHostMalloc(p, bytes, flags);
}
hipError_t raw_free(void *p, unsigned flags) { return flags == kDeviceMem ? hipFree(p) : hipHostFree(p); }

// empties the cache (of one device, or of all with device < 0); returns the bytes given back
size_t trim_locked(Pool &P, int device, std::vector<std::pair<void *, unsigned> > *out)
{
  size_t freed = 0;
  for (auto it = P.idle.begin(); it != P.idle.end();) {
    if (device >= 0 && it->first.device != device) { ++it; continue; }
    out->push_back({it->second, it->first.flags});
    freed += it->first.bytes;
    it = P.idle.erase(it);
  }
  P.cached -= freed;
  return freed;
}

hipError_t alloc(void **p, size_t bytes, unsigned flags)
{
  Pool &P = pool();
  if (!p) return hipErrorInvalidValue;
  if (!P.on || bytes == 0 || bytes > ((size_t)1 << 46)) return raw_alloc(p, bytes, flags);      // (an absurd size: the runtime's answer, no class arithmetic)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return raw_alloc(p, bytes, flags); }
  const size_t want = size_class(bytes);
  {
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.idle.lower_bound(Key{dev, flags, want});
    if (it != P.idle.end() && it->first.device == dev && it->first.flags == flags && it->first.bytes == want) {
      *p = it->second;
      P.live[*p] = it->first;
      P.cached -= it->first.bytes;
      P.idle.erase(it);
      ++P.hits;
      return hipSuccess;
    }
    ++P.misses;
    if (P.debug) std::fprintf(stderr, "[mvr pool] %zu bytes (class %zu, %s) from the runtime\n", bytes, want, flags == kDeviceMem ? "device" : "pinned");
  }
  hipError_t e = raw_alloc(p, want, flags);
  if (e != hipSuccess) {                       // out of memory with blocks lying idle: give them back and ask again
    (void)hipGetLastError();
    std::vector<std::pair<void *, unsigned> > gone;
    { std::lock_guard<std::mutex> lk(P.mu); (void)trim_locked(P, -1, &gone); }
    for (auto &g : gone) (void)raw_free(g.first, g.second);
    e = raw_alloc(p, want, flags);
  }
  if (e == hipSuccess) { std::lock_guard<std::mutex> lk(P.mu); P.live[*p] = Key{dev, flags, want}; }
  return e;
}

hipError_t release(void *p, bool host)
{
  if (!p) return hipSuccess;
  Pool &P = pool();
  Key k{0, 0, 0};
  bool keep = false, known = false;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.live.find(p);
    if (it != P.live.end()) {
      known = true; k = it->second;
      P.live.erase(it);
      keep = P.on && P.cached + k.bytes <= P.cap;
    }
  }
  if (!known) return host ? hipHostFree(p) : hipFree(p);        // (not from here: allocated while the pool was off)
  if (!keep) return raw_free(p, k.flags);
  // the block may still be in use by work queued on any stream of its device: drain it first (hipFree did that implicitly)
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  if (have_cur && cur != k.device) (void)hipSetDevice(k.device);
  const hipError_t e = hipDeviceSynchronize();
  if (have_cur && cur != k.device) (void)hipSetDevice(cur);
  if (e != hipSuccess) { (void)hipGetLastError(); return raw_free(p, k.flags); }      // (a broken device: nothing is kept from it)
  std::lock_guard<std::mutex> lk(P.mu);
  P.idle.insert({k, p});
  P.cached += k.bytes;
  return hipSuccess;
}
}  // namespace

hipError_t pool_malloc(void **p, size_t bytes) { return alloc(p, bytes, kDeviceMem); }
hipError_t pool_free(void *p) { return release(p, false); }
hipError_t pool_host_malloc(void **p, size_t bytes, unsigned flags) { return alloc(p, bytes, flags == kDeviceMem ? 0u : flags); }
hipError_t pool_host_free(void *p) { return release(p, true); }

}  // namespace mvr

extern "C" {
#define API __attribute__((visibility("default")))
// gives every idle block of the library's cache back to the runtime; returns the bytes freed.  stats (may be NULL): {bytes still
// cached, requests served from the cache, requests that went to the runtime}
API unsigned long long mvr_pool_trim(unsigned long long stats[3])
{
  mvr::Pool &P = mvr::pool();
  std::vector<std::pair<void *, unsigned> > gone;
  size_t freed = 0;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    freed = mvr::trim_locked(P, -1, &gone);
    if (stats) { stats[0] = P.cached; stats[1] = P.hits; stats[2] = P.misses; }
  }
  for (auto &g : gone) (void)mvr::raw_free(g.first, g.second);
  return (unsigned long long)freed;
}
}
