// csrc/mvr_internal.h -- shared declarations of libmvr_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "mvr_hip.h"

// ---- every device / pinned-host allocation of the library goes through a process-wide cache of freed blocks (mvr_pool.cpp): the names
// of the runtime's four calls are redirected HERE, after the runtime's and hipCUB's own headers (a translation unit that uses hipCUB
// includes it before this file), so that no call site has to know
namespace mvr {
hipError_t pool_malloc(void **p, size_t bytes);
hipError_t pool_free(void *p);
hipError_t pool_host_malloc(void **p, size_t bytes, unsigned flags);
hipError_t pool_host_free(void *p);
template <class T> inline hipError_t pool_malloc_t(T **p, size_t bytes) { return pool_malloc(reinterpret_cast<void **>(p), bytes); }
template <class T> inline hipError_t pool_host_malloc_t(T **p, size_t bytes, unsigned flags = 0) { return pool_host_malloc(reinterpret_cast<void **>(p), bytes, flags); }
template <class T> inline hipError_t pool_free_t(T *p) { return pool_free(const_cast<void *>(static_cast<const void *>(p))); }
template <class T> inline hipError_t pool_host_free_t(T *p) { return pool_host_free(const_cast<void *>(static_cast<const void *>(p))); }
}  // namespace mvr
#define hipMalloc(p, n) ::mvr::pool_malloc_t((p), (n))
#define hipFree(p) ::mvr::pool_free_t(p)
#define hipHostMalloc(...) ::mvr::pool_host_malloc_t(__VA_ARGS__)
#define hipHostFree(p) ::mvr::pool_host_free_t(p)

namespace mvr {

// ---- device data layout ------------------------------------------------------
// Morton ordering of one point SET, shared by every cloud that holds that set
// (a posed copy of a scan keeps the scan's ordering: rigid motion does not
// change which points are neighbours).
// Buffers of dropped orderings, kept for the next one: a target that grows by one scan per align gets a new
// ordering every time, and hipFree + hipMalloc of its two index arrays (a device-wide synchronisation each) cost
// more host time than the whole align took on the GPU (0.45 of 1.0 ms, rocprofv3 timeline of the sequential mode).
// One pool per context: a recycled buffer is next written by a kernel on the same stream, after its last reader.
struct OrderPool {
  struct Buf { uint32_t *p; size_t bytes; };
  std::vector<Buf> free;
  bool closed = false;
  uint32_t *take(size_t bytes, size_t *got);
  void give(uint32_t *p, size_t bytes);
  void close();
  ~OrderPool() { close(); }
};
struct Order {
  uint32_t *perm = nullptr;   // sorted position -> original index
  uint32_t *inv = nullptr;    // original index  -> sorted position
  size_t n = 0;
  size_t perm_bytes = 0, inv_bytes = 0;
  std::shared_ptr<OrderPool> pool;
  std::shared_ptr<char> arena;         // set: perm / inv are carved out of this block, shared with the other orderings of one batched build (freed with the last of them)
  ~Order();
};

// Uniform grid over a point SET in its CANONICAL coordinates (the coordinates it was uploaded with): like the ordering
// it belongs to the set, is built once per scan and shared by every posed copy -- a rigid motion moves the points, not
// their cells.  A search in a posed copy maps the query into the canonical frame (inverse of the copy's pose), walks the
// cells its search ball overlaps, and evaluates the candidates in the POSED frame (same float arithmetic as every other
// search kernel: bit-identical distances).  See mvr_grid.hip.
struct CellGrid {
  size_t n = 0;
  float lo[3] = {0, 0, 0};
  float h = 1.f, inv_h = 1.f;            // cell edge
  int dim[3] = {1, 1, 1};                // cells per axis; cell id = (z * dim[1] + y) * dim[0] + x
  uint32_t *start = nullptr;             // [cells + 1] first grid position of every cell (points sorted by cell id)
  // COMPACT form of the same table (round 4; `start` is then null): the cell ids are cut into segments of 32 consecutive ids; an
  // occupied segment has a record of its 32 starts in recs[], an empty one -- all its cells begin at the same position -- is the
  // directory entry itself: dir[seg] = slot | 0x80000000 or that position.  A 200k-point scan occupies ~1/20 of its 375k segments:
  // 1.5 MB of directory + 2.5 MB of records instead of 48 MB of which a walk touched a 128-byte line per row for three entries.
  uint32_t *dir = nullptr, *recs = nullptr;
  uint32_t nseg = 0;
  bool via_compact = false;              // a DENSE table that was built through the compact form and expanded (the default)
  int dt_shift = 0;                      // the distance map is kept per cube of 2^dt_shift cells per axis (compact form: 1 -- an eighth of the bytes and of the build)
  int dtdim[3] = {1, 1, 1};              // its dimensions
  uint32_t *gperm = nullptr;             // [n] grid position -> original index
  float4 *graw = nullptr;                // [n] canonical coordinates in grid order, w = bits(original index)
  uint32_t *g2h = nullptr;               // [n] grid position -> position in the set's Hilbert ordering (what the fused pass's keys carry)
  uint32_t *h2g = nullptr;               // [n] its inverse: where a seed (a Hilbert position) sits in grid order -- the seed's coordinates then come from the array the walk reads anyway
  char *block = nullptr;                 // the ONE device allocation the arrays below are carved out of
  hipEvent_t ready = nullptr;            // set when the grid was built on the context's side stream: recorded behind its last build kernel
  bool ready_waited = false;             // ... and the context's main stream has been made to wait for it
  uint8_t *dt = nullptr;                 // [cells] Chebyshev distance, in cells, to the nearest occupied cell (0 = occupied, 255 = farther than dt_steps): "is there any point near here at all" in ONE byte
  int dt_steps = 0;                      // dilation steps dt was built with: 255 means "farther than dt_steps cells"
  const Order *built_for = nullptr;      // the ordering g2h refers to
  ~CellGrid();
};

// A cloud is an array of float4 {x,y,z,1}: the same 16-byte record as
// pcl::PointXYZ (mvr/include/types.h:14), so a host upload is one memcpy and
// every lane moves one point with a single 16-byte access (dwordx4).
// For the culled search it also carries a Morton-ordered copy (w = bits of the
// original index) and one AABB per 256-point tile of that copy.
// A cloud may be a SHARD of a larger (distributed) cloud: segment k says that local points
// [local_begin, local_begin + count) are global points [global_begin, global_begin + count).
// No segments = the cloud is its own global numbering.
struct Seg { uint32_t local_begin, count, global_begin; };
constexpr int kMaxSegs = 64;
struct SegTable { uint32_t n; uint32_t lb[kMaxSegs], cnt[kMaxSegs], gb[kMaxSegs]; };

struct Mat44d { double m[16]; };
// pose-derived parameters of one posed cloud, in DEVICE memory: what the kernels of a pass that was enqueued before its
// poses were known read instead of by-value arguments (the pipelined ring run, mvr_ctx.hip)
struct PoseRec {
  Mat44d T;                 // the pose (column-major 4 x 4)
  double minv[12];          // its inverse affine map, row-major 3 x 4: posed frame -> canonical frame
  float stretch, delta;     // bound of how much the inverse lengthens a distance (1 for a rigid pose); how far (mm, upper bound) any point of the cloud has moved since the pass before (1e30: unknown)
};

// One merged scan inside a cloud that is a CONCATENATION of posed copies of scans -- the growing target of the sequential
// mode (`*target += transformed_source`, registrator.cpp:576).  Its points never move once appended, so the raw scan's
// pose-invariant cell grid serves it: the part records which point set it is a copy of and by which arithmetic its
// coordinates were made (kind 1: p = f32(pose * canonical) as mvr_cloud_transform; kind 2: that, then the f32 matrix of an
// align's result as mvr_icp_align's output) -- enough to write its coordinates in GRID order bit for bit and to map a
// query into the scan's canonical frame.
struct GridPart {
  unsigned long long set_id = 0; size_t n = 0, base = 0; int kind = 0;
  double pose[16]; float fin[16];
  std::shared_ptr<CellGrid> grid;      // found (or built) when the part is first searched
  bool gs_filled = false;              // its stretch of the cloud's gsorted[] is written
};

struct Cloud {
  float4 *pts = nullptr;
  size_t n = 0;
  size_t cap = 0;
  uint64_t set_id = 0;                 // identity of the point set + its indexing
  std::shared_ptr<Order> order;        // null until first needed
  float4 *sorted = nullptr; size_t sorted_cap = 0;
  float4 *tlo = nullptr, *thi = nullptr; size_t tiles_cap = 0;
  float4 *cbox = nullptr;              // [tiles][4 cells][lo, hi]: AABBs of the 64-point cells of every tile
  float4 *sbox = nullptr;              // [ceil(tiles / 64)][lo, hi]: AABBs of 64 consecutive tiles
  bool super_stale = false;            // sbox[] was NOT refreshed with the tile boxes (a posed copy about to be searched over its grid): flush_super_boxes() before a culled launch
  bool coords_valid = false;           // sorted[] / tlo / thi / cbox match pts[]
  size_t fresh_tiles = 0;              // with !coords_valid: that many LEADING tiles of sorted[] / boxes are still current (a cloud that only grew at its end: mvr_cloud_append with an extended ordering); any other change of the coordinates resets it
  void stale_coords() { coords_valid = false; fresh_tiles = 0; gcoords_valid = false; mgrid_ok = false; }
  // optional unit normals {nx,ny,nz,0} (point-to-plane extension, K10)
  float4 *nrm = nullptr; size_t nrm_cap = 0;
  bool has_normals = false;
  std::vector<Seg> segs;               // global numbering of a shard (target sharding over ranks)
  // pose bookkeeping for the grid search: `canonical` = these ARE the set's upload coordinates; otherwise, when
  // pose_known, pts = pose * (canonical coordinates of the set) exactly as mvr_cloud_transform computes it
  bool canonical = false, pose_known = false;
  double pose_stretch = 1.0;             // with pose_known: an upper bound of how much the INVERSE pose can lengthen a distance (1 for a rigid pose; the poses of a long registration are products of float-rounded matrices and drift away from orthonormal)
  double pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::shared_ptr<CellGrid> grid;          // the set's grid (shared)
  float4 *gsorted = nullptr; size_t gsorted_cap = 0;      // posed coordinates in grid order, w = bits(original index)
  bool gcoords_valid = false;          // gsorted matches pts
  std::shared_ptr<CellGrid> mgrid;     // a grid over THESE coordinates (the sequential mode's model searched as one point set: ensure_model_grid), valid while mgrid_ok
  bool mgrid_ok = false;
  // pipelined ring run (mvr_ctx.hip: ring_passes): where the kernels of a pass that is enqueued BEFORE its poses are known
  // find this posed copy's pose, inverse and stretch (device memory, filled by pose_prep_kernel once the host's solve has
  // released the pass); null outside such a run
  const struct PoseRec *pose_dev = nullptr;
  // ... or a known f64 pose FOLLOWED by a known f32 matrix (the output cloud of an align whose source had a known pose):
  // pts = xform_f32(fin, f32(pose * canonical)); the sequential mode's merged scans are made this way
  bool fin_known = false; float fin[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::vector<GridPart> parts;         // non-empty: this cloud is exactly the concatenation of these posed scans (see GridPart)
  void forget_pose() { canonical = false; pose_known = false; fin_known = false; pose_stretch = 1.0; grid.reset(); parts.clear(); }
  // how far this posed copy has moved since it was last searched (seed_delta, mvr_grid.hip): `moved` (mm, an upper bound over the
  // cloud's points; < 0: unknown) accumulates over the poses note_pose() records, from `last_pose`; a fused pass that searches the
  // cloud puts it back to 0.  Valid while last_pose_set == set_id.
  double last_pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, moved = -1.0; uint64_t last_pose_set = 0;
  float bbox[6] = {0, 0, 0, 0, 0, 0}; uint64_t bbox_set = 0;      // with bbox_set == set_id (and canonical): the bounding box of pts[], taken when the cloud was uploaded (the grid build then needs no round trip for it)
  bool pts_stale = false;              // pts[] were left out by the last (pipelined) pose of this cloud: sorted[] / gsorted[] are current, pts[] are written when the run leaves the pipe
  bool posed_by_table = false;         // ... and the last transform of this cloud did read it (its host-side pose is filled in when the run leaves the pipe)
};

constexpr uint32_t kNone = 0xFFFFFFFFu;      // "no neighbour" index
constexpr uint32_t kMarked = 0xFFFFFFFEu;    // slot[] state between CAS and list position

// NN key: (float bits of d2) << 32 | index.  d2 >= 0, so the unsigned order of
// the key is (d2, index) lexicographic: an integer min gives the nearest
// neighbour with ties resolved to the lowest index.
using nnkey_t = unsigned long long;
constexpr nnkey_t kKeyInit = ~0ull;

// brute-force NN tiling (see mvr_nn.hip)
constexpr int kNNThreads = 256;   // 4 waves
constexpr int kNNTile = 1024;     // target points per LDS tile (16 KB)
// Q (queries held in registers per lane) and SUB (sub-tile over which only
// min(d2) is tracked) are template parameters, chosen per ctx (nn_q, nn_sub).

// culled NN (see mvr_index.hip): tile of the Hilbert-ordered target cloud
constexpr int kCullTile = 256;
// its evaluation counters are sharded (one 128-byte line per shard) so that
// thousands of waves do not serialise on one address; region A = this launch,
// region B = {running total, max tiles per wave, max tiles tested} per shard
constexpr int kEvalShards = 64, kEvalStride = 16;
constexpr size_t kEvalRegion = (size_t)kEvalShards * kEvalStride;     // u64 per region
constexpr size_t kTraceBlocks = 4096, kTraceRec = 16;   // diagnostic builds (-DMVR_TRACE): per-block records of the last launch, after the two regions

// work = `work` unless `h_count` is set: then work = *h_count * per_count (the
// query count / evaluation count is only known on the device)
struct ProfRec { int family; hipEvent_t a, b; double work; const uint64_t *h_count; int n_shards; double per_count; };

// where the searches of one pair of the last fused batch left their keys (mvr_pair_batch_correspondences)
struct Ctx;
struct BatchPairRec { Ctx *w = nullptr; size_t off_s = 0, off_t = 0, qb = 0, qn = 0; int src = -1, dst = -1; double max2 = 0.0; int reciprocal = 0; unsigned long long src_set = 0, dst_set = 0; };

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 0;
  struct Cloud *refresh_rider = nullptr;              // refresh_posed_batch: a cloud whose stale index is refreshed by the same launch (set around a call, never kept)
  int reduce_rows = 0;                                // blocks (= partial rows) per pair of the launches that carry 29 sums per lane; 0: a quarter of the CUs (at least 32)
  int clock_mhz = 0;
  std::string name;
  std::string last_error;
  Cloud slots[MVR_MAX_SLOTS + 2];          // +2 internal scratch clouds
  uint64_t next_set_id = 1;
  std::map<uint64_t, std::weak_ptr<Order> > orders;   // set_id -> ordering (shared between posed copies)
  std::map<uint64_t, std::weak_ptr<CellGrid> > grids;     // set_id -> uniform grid (shared between posed copies)
  int grid_probe = 1;                                 // grid search: a query whose ball is wide first looks into the 2 x 2 x 2 cells nearest to it -- a point found there is a tighter (valid) bound, and most wide balls of a pass after a large motion are loose bounds, not far neighbours (0: off)
  int grid_probe_rows = 12;                           // ... the probe is made for balls of more than this many rows of cells (<= grid_light_rows; the 2 x 2 x 2 cells are four rows)
  int grid_light_rows_lone = 12;                      // ... in a launch of ONE pair (measured on the sequential mode's reverse searches: 24 costs an align 8 us)
  int grid_light_rows = 24;                           // grid search (default = kGridLightRows): rows of cells a thread walks itself; wider balls leave for a wave of their own or for the culled kernel
  // mvr_icp_align of a posed scan remembers, per point set, where each query's match sat in the target (sorted position) and
  // starts the NEXT align of that scan from the distance of that point under the current coordinates: the sweeps of the sequential
  // mode (registrator.cpp:530-577, repeat_times of them) and the rounds of AutoReg align the same scans again and again against a
  // model that has barely moved.  Any target point gives a valid inclusive bound, so a stale seed costs time, never exactness.
  int seq_seed = 1;
  struct SeedBuf { uint32_t *d = nullptr; size_t n = 0, cap = 0; };
  std::unordered_map<unsigned long long, SeedBuf> seq_seeds;      // by the source's point-set id
  uint32_t *seed_bound = nullptr; size_t seed_bound_cap = 0;
  struct G2HJob { std::shared_ptr<CellGrid> grid; std::shared_ptr<Order> order; size_t n; };
  std::vector<G2HJob> g2h_todo;                       // grid_coords_prepare -> flush_g2h
  int unseeded_grid = 0;                              // 1: the forward searches of a fused pass WITHOUT seeds (a registration's first) walk the grid as well (probe, then the listed sets) instead of the culled kernel
  int seq_model_tail = 1;                             // seq_search 3: the flagged query sets go 1 = to the culled kernel (listed sets), 0 = to the grid's set kernel
  int seq_cell_points = 4;                            // points per occupied cell the MODEL's grid aims at (seq_search 3)
  int seq_search = 1;                                 // mvr_icp_align of a posed scan: 1 (default) = the REVERSE searches walk the source scan's cell grid (compaction by the fused pass's kernels, no hipCUB), the forward search stays with the culled kernel; 2 = the forward search too, through the grids of the posed scans the target is made of (nn_parts_kernel: exact, measured slower -- DESIGN.md 4.5); 0 = the culled kernel for both
  int parts_lanes = 0, parts_max_rows = 25;           // nn_parts_kernel: lanes per query (1, 2, 4, 8; 0 = by the number of parts) and the widest ball (rows of cells) a lane walks itself
  int ring_search = 1;                                // fused pass: 1 = seeded searches walk the uniform grid (thread per query), 0 = always the culled kernel
  std::shared_ptr<OrderPool> order_pool = std::make_shared<OrderPool>();
  // per-pair work buffers (grown on demand)
  nnkey_t *keys = nullptr;   size_t keys_cap = 0;     // [Ns] forward NN keys (by original source index)
  nnkey_t *rkeys = nullptr;  size_t rkeys_cap = 0;    // [Nt'] reverse NN keys (by list position)
  uint32_t *slot = nullptr;  size_t slot_cap = 0;     // [Nt] target -> list position
  uint32_t *list = nullptr;  size_t list_cap = 0;     // [Nt'] distinct matched targets
  int32_t *match = nullptr;  size_t match_cap = 0;    // [Ns] accepted match or -1
  uint8_t *flags = nullptr;  size_t flags_cap = 0;    // [Nt] matched-target flags (culled mode, sorted space)
  uint32_t *bound = nullptr; size_t bound_cap = 0;    // [Nt] start bounds of the reverse searches (culled mode, sorted space)
  uint32_t *count = nullptr;                          // device counter (list length)
  unsigned long long *evals = nullptr;                // device counter: pair evaluations of the culled kernel
  double *partials = nullptr; size_t partials_cap = 0;
  double *moments = nullptr;                          // device: 64 doubles
  double *h_moments = nullptr;                        // pinned host: 64 doubles
  // mvr_icp_align's one round trip per iteration: the last sums launch stores the iteration's 19 doubles into mapped pinned
  // memory itself and then a sequence word the host spins on (align_spin; a copy packet + hipStreamSynchronize cost ~25 us)
  double *h_align = nullptr, *d_align = nullptr;      // pinned host, 64 doubles + the word (at double 64), and the device's view of it
  uint32_t align_seq = 0;
  int align_spin = 1;
  int seq_rider = 1;                                  // mvr_seq_run: the model's tail is refreshed by the launch that poses the next source
  // index-build scratch
  uint32_t *codes_a = nullptr, *codes_b = nullptr, *idx_a = nullptr; size_t sort_cap = 0;
  void *cub_tmp = nullptr; size_t cub_cap = 0;
  struct PartDesc *d_parts = nullptr, *h_parts = nullptr; size_t parts_cap = 0;      // the part table of a composite target: device copy and pinned staging
  char *scratch = nullptr; size_t scratch_cap = 0;    // temporaries of the grid builds (grows, never shrinks)
  char *oscratch = nullptr; size_t oscratch_cap = 0;  // temporaries of a batched ordering build (its own buffer: the grid builds use `scratch` on the side stream meanwhile)
  int order_batch = 1;                                // 1 (default): the orderings a call needs for several point sets are built together (one sort); 0: one set after the other
  hipStream_t side_stream = nullptr;                  // set-up work that may overlap a pass (grid builds)
  hipEvent_t side_after = nullptr;                    // recorded on the main stream before the pass the side stream's work overlaps
  hipEvent_t scratch_event = nullptr; hipStream_t scratch_stream = nullptr;      // the last user of `scratch`
  float *bbox = nullptr;                              // device: 8 floats (lo xyz, hi xyz)
  // launch configuration (mvr_ctx_tune)
  int nn_q = 8, nn_sub = 32, nn_blocks_per_cu = 2;
  int nn_mode = 1;                                    // 0: brute force, 1: culled (exact, identical results)
  int cull_q = 0;                                     // culled kernel: queries per lane (0 = auto)
  int cull_w = 0;                                     // culled kernel: waves sharing one query set (1, 2, 4; 0 = by launch size)
  int cull_slices = 0;                                // culled kernel: interleaved slices a pair's query sets are dealt to the XCDs in (1, 2, 4, 8; 0 = 8 / gcd(pairs, 8))
  // seed_delta (round 4): a forward search of the grid walk that has the previous pass's key at hand takes its start bound from THAT
  // distance plus how far the two clouds have moved since (an upper bound from the poses and the scan's bounding box) instead of
  // gathering the old match's coordinates (two dependent gathers at the head of every wave) -- while the motion is below
  // seed_delta_um micrometres (0: never, the default).  Any bound within which a point is known to exist is a valid start: results do
  // not change (tests/test_gpu_ring.py) -- and, measured, neither does the time: 0.339 / 0.329 ms per settled pass with it against
  // 0.336 / 0.330 without (same box, 50 um): the two gathers it removes are latency other waves cover.  Kept as a knob; the
  // per-view motion bound it rests on (PoseRec::delta, Cloud::moved) is what a certificate for the rim queries would need too.
  int seed_delta_um = 0;
  // rim certificates: a forward query without any target point within the cap + rim_cert_um micrometres (found so by the listed-sets
  // search) is not searched again -- no distance-map read, no probe, no place in a listed set -- while the pair has moved by less
  // than that since (the same per-view motion bounds; 0: off, the default).  Results do not change (knob tests) -- and neither does
  // the time: of the ~30 k rim queries per 1.2 M of a settled pass, 50 / 200 / 500 / 1000 um of margin take 0 / 2 / 8 / 9 k out and the
  // listed SETS (a block each: what the tail launch costs) go from 950 to 800 at best -- a rim query is not a query with nothing
  // near, it is one whose nearest target point lies just beyond the cap, and a registration that still moves 0.02 mm per view and
  // pass spends a small margin at once.  The tail launch stays at 22-24 us either way (profiles/r04_c_rim_cert.txt).
  int rim_cert_um = 0;
  float *bcert = nullptr; size_t bcert_cap = 0; bool bcert_zero = false; float bcert_cap2 = -1.f;      // (the cap the certificates in bcert[] were granted for)
  int seed_forward = 1;                               // fused pass: forward searches start from the previous pass's matches when the same pairs are searched again
  int fused_mark = 1;                                 // fused pass: the forward launches themselves record the matched targets' start bounds -- 1: when they are the grid walk, 2: always, 0: never (a separate launch re-reads the keys)
  bool marked_in_search = false;                      // ... what this pass does (decided with its forward launches)
  std::vector<BatchPairRec> last_batch;               // the pairs of the last fused mvr_pair_moments2_batch on this context
  unsigned long long fused_passes = 0;                // fused pair batches run on this context so far
  std::vector<unsigned long long> fused_sig;          // what the forward keys in bkeys[] belong to (point-set ids, ranges, offsets): the previous fused pass on this context
  int pair_streams = 6;                               // worker streams of mvr_pair_moments2_batch
  // workers: contexts with their own stream and work buffers that BORROW clouds of this
  // context (slots 0/1) so that independent scan pairs run concurrently on the GPU
  std::vector<Ctx *> workers;
  Ctx *parent = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  double *batch_table = nullptr; size_t batch_cap = 0;   // [pairs][32] staging of mvr_pair_moments2_batch
  char *dn_arena = nullptr; size_t dn_arena_cap = 0;     // scratch of mvr_cloud_denoise (one buffer, carved per call)
  double *h_table = nullptr; size_t h_table_cap = 0;     // pinned host memory the final kernels of mvr_ring_step write straight into (zero-copy)
  // workspace of the fused batch (culled mode): forward keys of all pairs, reverse keys + flags, per-pair partial rows
  nnkey_t *bkeys = nullptr; size_t bkeys_cap = 0;
  nnkey_t *brkeys = nullptr; size_t brkeys_cap = 0;
  uint32_t *bwide = nullptr; size_t bwide_cap = 0;       // [sources + targets of all pairs] grid search: ordinals of the wide bounded queries
  int grid_cell_points = 4;                              // points per occupied cell the grid's cell edge aims at (grids built from then on)
  int grid_index = 2;                                    // a grid's cell-start table (grids built from then on): 2 (default) = DENSE, built through the compact form (all scans sorted at once, distance map per 2 x 2 x 2 cells) and expanded; 1 = COMPACT (segment directory + records of the occupied segments: a tenth of the bytes, walks 8-11 % slower); 0 = dense by the round-3 build (a sort, a scan over all cells and a per-cell distance map per grid)
  // the STAGED walk (mvr_grid.hip): a wave of the one-lane-per-query walk copies the box of its lanes' cells into LDS with a few
  // coalesced loads and every lane walks its own cells there -- the walk is bound by the number of vector memory instructions a
  // wave issues, not by bytes.  0: the plain walk; 1 (default): launches over a scan's own query order (the forward searches); 2: every
  // launch, compacted query lists too (measured equal or slower there).  A lone pair's launch has its own switch.
  int grid_stage = 2, grid_stage_lone = 1;      // (2 since the two waves of a block stage one region together: the reverse launches' sparser queries fit it too)
  unsigned long long *stage_stat = nullptr;              // diagnostics (tune key grid_stage_stat = 1): device counters [64 shards][64] of the staged walk's waves: [1] staged, [2] too many rows, [3] too wide, [4] too many points
  int grid_lanes = 1;                                    // lanes that share one query of the grid search (1, 2, 4 or 8): they deal the ball's rows of cells among them
  int grid_cluster = 8;                                  // a wave of the grid search with at least this many wide queries hands them ALL to the culled kernel (65: never)
  int grid_wide_waves = 32;                              // waves per CU of the wave-per-query launch
  int grid_tail = 1;                                     // 1: the wave-per-query launch and the listed-sets launch of a forward pass are one launch
  int grid_sets = 1;                                     // the listed query sets of a seeded forward pass: 1 = over the grid (nn_grid_set_kernel) while a set's union of balls is a few hundred rows of cells, 2 = always over the grid, 0 = by the culled kernel over the set list
  int cull_list = 1;                                     // 1: the grid search lists the query sets it flags and the culled kernel walks that list (0: a block per set, most of which leave at once)
  uint32_t *bcull_sets = nullptr; size_t bcull_sets_cap = 0;
  int cull_list_w = 2;                                   // waves per query set of the set-list launch of the culled kernel (1, 2 or 4)
  int grid_debug = 0;                                    // 1: every fused pass prints how its queries split between the three searches (synchronises: diagnostics only)
  int grid_wide = 1;                                     // 1: wide BOUNDED queries of the grid search get a wave each (0: flagged for the culled kernel like the unbounded ones)
  uint32_t *bwide_count = nullptr;                       // [2 * kBatchPairs ...] their counts (zeroed by the pass's moments launch)
  uint8_t *bheavy = nullptr; size_t bheavy_cap = 0;      // [sources of all pairs] grid search: queries left to the culled kernel (wide balls)
  uint32_t *bbound = nullptr; size_t bbound_cap = 0;     // [targets of all pairs] bits of the forward d2 of a source that matched the target (~0: not matched)
  bool bbound_clean = false;                             // bbound[] is all ~0 (the moments launch of a pass restores what its forward launch marked)
  uint32_t *blist = nullptr; size_t blist_cap = 0;
  uint32_t *bslot = nullptr; size_t bslot_cap = 0;
  uint32_t *bchunks = nullptr; size_t bchunks_cap = 0;      // per-chunk counts / offsets, then one count per pair
  double *bpartials = nullptr; size_t bpartials_cap = 0;
  int inplace_ratio = 3;                              // culled reverse search in place (flags) when nt < ratio * queries, else compacted list
  int pair_fused = 1;                                 // culled mode: all pairs of a batch in one launch per stage (0: worker streams)
  int posed_refresh = 1;                              // mvr_cloud_transform_batch brings the posed copies' index up to date from the sources' sorted copies
  int pair_groups = 2;                                // fused pass: groups of pairs on concurrent streams (1: a single stream); measured on the 12-pair ring: 1.27 / 1.22 / 1.28 / 1.39 ms per step with 1 / 2 / 3 / 4
  // ---- pipelined ring run (ring_passes, mvr_ctx.hip): pass k+1's whole launch chain is enqueued while pass k runs, behind
  // a gate the host opens after its solve; the poses reach the kernels through a device table instead of kernel arguments
  int pipeline = 1;                                   // 0: every pass is enqueued after the previous solve (round-2 behaviour)
  int pipeline_multi_rank = 0;                        // 1: passes with a collective over more than one rank may be queued behind the gate too (off until a recorded two-GPU run says so)
  bool no_sync = false;                               // a gated chain is being enqueued: nothing may wait for the stream or reallocate (may_block)
  unsigned long long blocking_events = 0;             // how often something did wait / reallocate (a pass without any is in steady state)
  bool single_group = false;                          // the fused pass runs its pairs as ONE group for the whole ring run: a run that may queue passes ahead must not change the layout of its work buffers (and of the seeds in them) when it starts to
  bool skip_posed_pts = false;                        // ... and the posed copies' points in ORIGINAL order are not written (nothing in a fused pass reads them: ring_passes writes them once, when the stretch of queued passes ends)
  bool pose_from_table = false;                       // mvr_cloud_transform_batch ignores its T values: every destination reads Cloud::pose_dev
  double *h_pose_in = nullptr, *d_pose_in = nullptr;  // pinned, mapped: [2][views][16] poses the host writes before it opens the gate
  PoseRec *pose_tab = nullptr; size_t pose_tab_cap = 0;      // device: [2][views]
  // the chain being enqueued reads its poses from pose_in_cur (pinned) / pose_tab_cur (device records, n of them); pending: the records
  // are not filled yet -- the posing launch of the chain does it on the way (refresh_batch), any other reader calls ensure_pose_table first
  // the completion word of the chain being enqueued: stored by the final sums launch itself when that launch is the last thing
  // of the chain (signal_armed, set by the pass loop for hosts that end a pass with the fused sums on this stream; signal_sent
  // tells the loop that it needs no write-value operation behind the chain)
  uint32_t *done_counter = nullptr; uint32_t signal_seq = 0; bool signal_armed = false, signal_sent = false;
  int lazy_super = 1;                // 1 (default): posed copies that get grid-ordered coordinates leave their super boxes (read by the culled kernel alone) to flush_super_boxes(); 0: refreshed with every posing launch
  int setup_first = 1;               // plain passes: 1 (default) = the grid builds (side stream) and the pipe's set-up are enqueued BEFORE the pass's own chain instead of behind it: the first pass of a registration is host-bound either way, and the grids are then ready when the second pass wants them (12 x 200k: 2.8 + 2.1 ms -> 3.7 + 0.5)
  int pose_prep_launch = 0;          // test hook (tune key): 1 = the device pose records are always filled by the launch made for that, never by the posing launch on the way
  const double *pose_in_cur = nullptr; PoseRec *pose_tab_cur = nullptr; int pose_tab_n = 0; bool pose_tab_pending = false;
  uint32_t *gate = nullptr; bool gate_is_signal = false;     // host-writable word the stream waits on (hipStreamWaitValue32)
  uint32_t *h_done = nullptr, *d_done = nullptr;      // pinned, mapped: the stream writes the pass number here when a pass's chain has drained
  bool pipe_ops_warm = false;                         // the stream's wait-value / write-value operations have been used once
  uint32_t pipe_seq = 0;                              // passes sent through the pipe so far
  unsigned long long pipe_steady_sig = 0, pipe_steady_events = ~0ull;      // the registration whose last run ended in steady state, and blocking_events then
  std::vector<double> pass_ms;                        // wall time of every pass of the last ring run (end of the previous solve -> end of this one)
  unsigned long long piped_passes = 0;                // passes of this context whose chain was enqueued ahead of their poses (diagnostics)
  // multi-GPU (mvr_world.cpp): an RCCL communicator (ncclComm_t; null = this context is a world of its own)
  void *comm = nullptr; bool comm_owned = false; int comm_rank = 0, comm_world = 1;
  void **comm_lender = nullptr;                       // a world's communicator is lent: where the world keeps it (cleared when it is aborted here)
  bool stream_stuck = false;                          // ncclCommAbort did not return: the stream may never drain (waits on it are bounded from then on)
  bool comm_broken = false;                           // the communicator was aborted (a peer failed or never arrived): the multi-GPU entry points refuse until a new one is set
  // PROJECTION of a rank's share on one GPU (mvr_ctx_project; tools/rank_share_bench.py): mvr_ring_run_sharded then plans as
  // rank project_rank of project_world whatever the communicator says (a world of one: the collective is really issued, its
  // peers are not there), and the rows the absent peers would have contributed are added to the table behind the all-reduce
  int project_world = 0, project_rank = 0;
  double *proj_rows = nullptr; size_t proj_rows_cap = 0, proj_rows_n = 0;
  int wait_timeout_ms = 60000;                        // how long a rank waits for a pass that contains a collective before it declares its peers lost
  // failure injection (tests; mvr_ctx_tune): the dist_pass-th sharded pass / iteration since the knob was set
  long long dist_pass = 0, inject_fail_at = -1, inject_stall_at = -1;
  uint32_t *h_stall = nullptr, *d_stall = nullptr;    // pinned word an injected stall waits on (released by comm_abort)
  long long *seq_keys = nullptr; size_t seq_keys_cap = 0;    // [Ns] signed keys with global target indices (sequential mode, target sharded)
  double *seq_row = nullptr;                                 // device: 40 doubles = one row of sums + the failure count of the iteration
  double *dist_table = nullptr; size_t dist_table_cap = 0;   // [edges][32]: this rank's rows, all-reduced in place
  // instrumentation
  bool prof = false;
  unsigned prof_mask = ~0u;                           // families that are timed (bit f = family f)
  bool prof_totals = false;                           // level 2: culled evaluations accumulate in region A, read once at drain
  std::vector<hipEvent_t> event_pool;                 // timing events are recycled, not re-created per launch
  std::vector<ProfRec> recs;
  uint64_t *h_counts = nullptr;                       // pinned: device counters copied per profiled launch
  uint64_t *h_evals = nullptr;                        // pinned [kEvalRegion]: the culled / grid kernels' running totals, copied with every iteration's moments (mvr_icp_align's `evals` statistic without a wait of its own)
  size_t h_counts_used = 0;
  uint64_t prof_launches[MVR_K_COUNT] = {0};
  double prof_ms[MVR_K_COUNT] = {0};
  double prof_work[MVR_K_COUNT] = {0};
};

constexpr int kScratchCur = MVR_MAX_SLOTS;       // ICP's input_transformed
constexpr int kScratchTmp = MVR_MAX_SLOTS + 1;   // fitness temporary

int set_error(Ctx *c, int status, const char *what, hipError_t e = hipSuccess);
// host-side stopwatch for the set-up phases (MVR_TRACE_HOST=1): prints the milliseconds since the previous mark to stderr
void host_mark(const char *what);

#define MVR_HIP_TRY(ctx, expr)                                               \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) return ::mvr::set_error((ctx), MVR_E_HIP, #expr, _e); \
  } while (0)

constexpr size_t kProfCounts = 1 << 20;   // pinned u64 slots (a culled launch takes kEvalRegion of them)

struct ProfScope {
  Ctx *c; int fam; hipEvent_t a = nullptr, b = nullptr; double work;
  const void *d_count = nullptr; int d_bytes = 0; int n_shards = 0; double per_count = 0.0;
  ProfScope(Ctx *ctx, int family, double w);
  // work is (*dev_count) * per_cnt, read back asynchronously after the launch;
  // dev_count points to a device uint32 (bytes = 4), a uint64 (bytes = 8), or,
  // with shards > 0, to `shards` u64 counters kEvalStride apart that are summed
  ProfScope(Ctx *ctx, int family, const void *dev_count, int bytes, double per_cnt, double upper_bound, int shards = 0);
  ProfScope(const ProfScope &) = delete;
  ProfScope &operator=(const ProfScope &) = delete;
  ~ProfScope();
};

// A pass enqueued behind a host-released gate must not wait for its own stream (the host would wait for itself) nor free
// what queued kernels use: every place that synchronises or reallocates asks here first.  The runner treats the error as
// "this pass is not in steady state yet" and enqueues it the ordinary way.
inline int may_block(Ctx *c, const char *what)
{
  Ctx *top = c->parent ? c->parent : c;
  ++top->blocking_events;
  if (top->no_sync) return set_error(c, MVR_E_HIP, what);
  return MVR_OK;
}
#define MVR_MAY_BLOCK(ctx, what) do { if (int rc_ = ::mvr::may_block((ctx), "not in steady state: " what)) return rc_; } while (0)

template <class T>
int ensure(Ctx *c, T *&p, size_t &cap, size_t want)
{
  if (cap >= want) return MVR_OK;
  MVR_MAY_BLOCK(c, "a work buffer has to grow");
  if (p) { (void)hipStreamSynchronize(c->stream); (void)hipFree(p); p = nullptr; cap = 0; }
  size_t ncap = want + want / 4 + 64;
  MVR_HIP_TRY(c, hipMalloc(&p, ncap * sizeof(T)));
  cap = ncap;
  return MVR_OK;
}

// several clouds per launch (kernel arguments by value)
constexpr int kBatchClouds = 16;
// (Tp[k], when set, overrides T[k]: the pose is read from device memory -- see PoseRec)
struct XformBatch { const float4 *src[kBatchClouds]; float4 *dst[kBatchClouds]; unsigned long long n[kBatchClouds]; Mat44d T[kBatchClouds]; const Mat44d *Tp[kBatchClouds]; };
// p' = T p as mvr_cloud_transform (f64 pose, f32 result): the ONE definition, shared by the transform kernels
// and by the index refresh that poses a scan's sorted copy directly (same operations, same bits)
__device__ __forceinline__ float4 pose_point_f64(const Mat44d &T, const float4 p)
{
  const double x = p.x, y = p.y, z = p.z;
  const double d = 1.0 / (((T.m[3] * x + T.m[7] * y) + T.m[11] * z) + T.m[15]);
  float4 o;
  o.x = (float)((((T.m[0] * x + T.m[4] * y) + T.m[8] * z) + T.m[12]) * d);
  o.y = (float)((((T.m[1] * x + T.m[5] * y) + T.m[9] * z) + T.m[13]) * d);
  o.z = (float)((((T.m[2] * x + T.m[6] * y) + T.m[10] * z) + T.m[14]) * d);
  o.w = 1.0f;
  return o;
}
// the device record of a pose (PoseRec) from its 16 doubles: inverse of x -> A x + t by cofactors, in double, as make_grid_pair
// does on the host; the stretch for ANY invertible matrix (the host has checked e <= 1e-3 before it released the pass: note_pose's bar)
__device__ __forceinline__ void make_pose_rec(const double *T, float delta, PoseRec *out)
{
  PoseRec r;
  for (int j = 0; j < 16; ++j) r.T.m[j] = T[j];
  const double A[3][3] = {{T[0], T[4], T[8]}, {T[1], T[5], T[9]}, {T[2], T[6], T[10]}};
  const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                     A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  const double id = 1.0 / det;
  const double I[3][3] = {{(A[1][1] * A[2][2] - A[1][2] * A[2][1]) * id, (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id, (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id},
                          {(A[1][2] * A[2][0] - A[1][0] * A[2][2]) * id, (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id, (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id},
                          {(A[1][0] * A[2][1] - A[1][1] * A[2][0]) * id, (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id, (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id}};
  for (int rr = 0; rr < 3; ++rr) {
    for (int k = 0; k < 3; ++k) r.minv[4 * rr + k] = I[rr][k];
    r.minv[4 * rr + 3] = -(I[rr][0] * T[12] + I[rr][1] * T[13] + I[rr][2] * T[14]);
  }
  double e2 = 0.0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      const double d = T[4 * a] * T[4 * b] + T[4 * a + 1] * T[4 * b + 1] + T[4 * a + 2] * T[4 * b + 2];
      const double x = d - (a == b ? 1.0 : 0.0);
      e2 += x * x;
    }
  const double e = fmin(sqrt(e2), 0.5);
  r.stretch = __double2float_ru((1.0 / sqrt(1.0 - e)) * (1.0 + 1e-6));
  r.delta = delta;
  *out = r;
}
struct RefreshBatch {
  const float4 *pts[kBatchClouds]; const uint32_t *perm[kBatchClouds]; unsigned long long n[kBatchClouds];
  float4 *sorted[kBatchClouds], *tlo[kBatchClouds], *thi[kBatchClouds], *cbox[kBatchClouds], *sbox[kBatchClouds];
  // optional: the sorted copy of the cloud this one is a posed copy of (same ordering) and the pose -- the refresh
  // then reads that copy in order and poses it, instead of gathering pts[] through the permutation
  const float4 *from[kBatchClouds]; Mat44d T[kBatchClouds]; const Mat44d *Tp[kBatchClouds];      // (Tp[k], when set, overrides T[k])
  // optional, with `from`: the source's points in original order and where their posed copies go (the transform itself,
  // done by the same launch)
  const float4 *xsrc[kBatchClouds]; float4 *xdst[kBatchClouds];
  // optional, with `from`: the cloud's points in the order of its set's cell grid (canonical coordinates) and where their posed
  // copies go -- the grid search's side of a posed copy, written by the same launch (else: refresh_grid_coords_batch)
  const float4 *graw[kBatchClouds]; float4 *gout[kBatchClouds];
  // optional, with Tp: the pose's 16 doubles where the host put them (pinned memory) -- every block reads them from THERE and block 0
  // of the cloud fills the device record Tp[k] points into (PoseRec) for the launches that follow: no launch of its own for that
  const double *Tin[kBatchClouds];
  unsigned tile_begin[kBatchClouds];       // refresh tiles from this one on (a cloud that grew at its end keeps its leading tiles)
};
static_assert(sizeof(RefreshBatch) <= 4096, "RefreshBatch travels as a kernel argument");
// index of posed copies, straight from the sources' sorted copies (culled mode; called by mvr_cloud_transform_batch).
// with_pts: also write dst->pts = T * src->pts.  handled[k] (optional) = cloud k was refreshed by this call.
int refresh_posed_batch(Ctx *c, int count, Cloud *const *dst, Cloud *const *src, const double *T, bool with_pts = false,
                        char *handled = nullptr);
int launch_transform_f64_batch(Ctx *c, int count, const float4 *const *in, float4 *const *out, const size_t *n, const double *T,
                               const Mat44d *const *Tp = nullptr);
int ensure_index_batch(Ctx *c, Cloud *const *clouds, int count);     // ensure_index for many clouds, coordinates refreshed in one launch

// ---- kernel launchers (mvr_nn.hip / mvr_reduce.hip / mvr_index.hip) -----------
// forward / reverse brute-force NN.  Queries: points [q_begin, q_begin+q_count)
// of `q` (direct; key ordinal = point index), or, when `qlist` != null, the
// points q[qlist[k]] for k < *qcount (device counter; q_count is then the upper
// bound used to size the grid; key ordinal = k).
int launch_nn(Ctx *c, const float4 *q, size_t q_begin, size_t q_count, const uint32_t *qlist,
              const uint32_t *qcount, const float4 *t, size_t nt, bool fma, nnkey_t *keys);

// Morton order + tile AABBs of a cloud, (re)built or refreshed as needed
int ensure_index(Ctx *c, Cloud &cl);
int flush_super_boxes(Ctx *c);                         // the super boxes the posing launches left out (lazy_super), for every cloud of the context: called by the culled launchers
void new_point_set(Ctx *c, Cloud &cl);                 // after upload / append / clear
void inherit_point_set(Cloud &dst, const Cloud &src);  // after copy / transform
// dst has just grown from old_n points by the points of `src` (appended at its end): keep dst's ordering and extend it by
// src's own, shifted -- the merged target of the sequential mode then never sorts its 2M points again.  false: not
// applicable (no ordering on either side, a shared ordering, no room): the caller falls back to new_point_set.
bool extend_point_set(Ctx *c, Cloud &dst, size_t old_n, const Cloud &src);
// culled exact NN.  Queries: sorted positions [q_begin, q_begin+q_count) of the query cloud; key ordinal = ORIGINAL
// index of the query, or -- with qflags (one byte per sorted position, only flagged positions are searched) -- the
// sorted position.  cap2: distances above it are not needed (+inf = unbounded).  The batch form runs the searches
// of several scan pairs in ONE launch (blockIdx.y = pair).
constexpr int kBatchPairs = 16;
struct CullPair {
  const float4 *qs = nullptr, *ts = nullptr, *tlo = nullptr, *thi = nullptr, *cbox = nullptr, *sbox = nullptr;
  const uint8_t *qflags = nullptr;
  const uint32_t *qlist = nullptr, *qcount = nullptr;     // compacted queries (sorted positions) and their device count
  nnkey_t *keys = nullptr;
  uint32_t q_begin = 0, q_count = 0, nt = 0, n_tiles = 0;
  const uint32_t *qbound = nullptr;   // optional, by sorted position: bits of a distance (squared) within which the query is KNOWN to have a point -- the search starts from that bound instead of the cap
  uint32_t *mark = nullptr;      // optional, with key_by_pos: [nt] start bounds of the reverse searches, by the target's sorted position (all ~0 on entry).  A query that finds a match within the cap stores the bits of its d2 at its match's position -- what a separate "flag the matched targets" launch did by re-reading all the keys.  Any matching source's distance is a valid bound: relaxed stores, last one wins.
  uint32_t key_by_pos = 0;       // plain queries only: key slot = the query's sorted position (coalesced stores) instead of its original index, AND the key's low word = the match's sorted position instead of its original index
  uint32_t seed_from_keys = 0;   // with key_by_pos: keys[] still holds the previous result of the same queries against the same target point set; every search starts from the distance of its previous match
  const uint32_t *setlist = nullptr, *setcount = nullptr;     // with qflags: the query sets that HOLD a flagged query (written by the launch that set the flags) -- launch_nn_cull_list_batch walks these instead of starting a block per set
};
struct CullBatch { CullPair p[kBatchPairs]; float cap2; };
CullPair make_cull_pair(const Cloud &q, size_t q_begin, size_t q_count, const uint8_t *qflags, const Cloud &t, nnkey_t *keys);
// block -> (pair, query set) of a fused launch, XCD-aware (see mvr_cull.hip: nn_cull_kernel)
struct XcdMap { uint32_t sets[kBatchPairs]; uint32_t n_pairs, slices; };
__host__ __device__ inline uint32_t slice_len(uint32_t sets, uint32_t slices, uint32_t slice)
{
  return sets > slice ? (sets - slice + slices - 1u) / slices : 0u;
}
__host__ __device__ inline bool xcd_map_block(const XcdMap &m, uint32_t block, uint32_t *pair, uint32_t *set)
{
  uint32_t r = block >> 3;
  for (uint32_t u = block & 7u; u < m.n_pairs * m.slices; u += 8u) {
    const uint32_t p = u / m.slices, sl = u % m.slices, len = slice_len(m.sets[p], m.slices, sl);
    if (r < len) { *pair = p; *set = sl + r * m.slices; return true; }
    r -= len;
  }
  return false;
}

inline void xcd_map_plan(XcdMap &map, int forced_slices, unsigned *grid_blocks)
{
  uint32_t g = 8u;
  while (map.n_pairs % g) g >>= 1;                            // gcd(pairs, 8)
  map.slices = 8u / g;
  if (forced_slices == 1 || forced_slices == 2 || forced_slices == 4 || forced_slices == 8) map.slices = (uint32_t)forced_slices;
  uint32_t per_xcd = 0;                                       // the longest of the 8 block lists
  for (uint32_t v = 0; v < 8u; ++v) {
    uint32_t tot = 0;
    for (uint32_t u = v; u < map.n_pairs * map.slices; u += 8u) tot += slice_len(map.sets[u / map.slices], map.slices, u % map.slices);
    per_xcd = per_xcd > tot ? per_xcd : tot;
  }
  *grid_blocks = 8u * per_xcd;
}
int launch_nn_cull_batch(Ctx *c, const CullPair *pairs, int n_pairs, float cap2, bool fma);
int launch_nn_cull_list_batch(Ctx *c, const CullPair *pairs, int n_pairs, float cap2, bool fma);     // only the query sets on the pairs' set lists (Q = 1)
// ---- grid search (mvr_grid.hip): exact 1-NN of seeded / bounded queries, one thread per query
struct GridPair {
  const float4 *qs = nullptr;                 // queries: a Hilbert-ordered posed cloud (w = original index)
  const uint32_t *qlist = nullptr, *qcount = nullptr;     // optional: compacted query positions + their device count (key slot = list position)
  uint32_t q_begin = 0, q_count = 0;
  const float4 *gts = nullptr;                // target: posed coordinates in GRID order (w = original index)
  const float4 *ts = nullptr;                 // target in Hilbert order (only to price the seed: the previous match's position)
  const uint32_t *start = nullptr, *g2h = nullptr, *h2g = nullptr, *tinv = nullptr;      // (tinv: original index -> Hilbert position in the searched cloud)
  const uint32_t *dir = nullptr, *recs = nullptr;      // the compact form of the cell starts (CellGrid): used when dir is set
  int dt_shift = 0, dtdim[3] = {1, 1, 1};
  const uint8_t *dt = nullptr;
  float lo[3] = {0, 0, 0}, inv_h = 1.f, h = 1.f;
  int dim[3] = {1, 1, 1};
  uint8_t *heavy = nullptr;                   // optional, by query position (by LIST position with qlist): a query whose ball spans more than `light_rows` rows of cells is NOT answered here but flagged (1, else 0) for the culled kernel, which is built for wide searches
  double minv[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};     // posed frame -> canonical frame of the target: r = minv (3 x 4, row-major) * (p, 1)
  uint32_t nt = 0;
  nnkey_t *keys = nullptr;
  const uint32_t *qbound = nullptr;           // optional start bounds by query position (bits of a d2; ~0 = none)
  uint32_t *mark = nullptr;                   // optional: start bounds of the reverse searches, by the match's Hilbert position
  uint32_t key_by_pos = 0, seed_from_keys = 0;
  const PoseRec *pose_dev = nullptr;     // optional: minv / stretch are read from here (device-visible) instead of the two by-value members
  const PoseRec *qpose_dev = nullptr;    // ... and the QUERY cloud's record (for its `delta`)
  // rim certificates (round 4): per plain forward query (by position), how much more (mm) the pair may move before the finding "no
  // target point within the cap" of an earlier pass has to be proved again; 0 = none.  Granted by the listed-sets search (which then
  // looks cert_margin beyond the cap), spent by the walk (GridBatch::cert_margin; delta as for seed_delta)
  float *cert = nullptr;
  float delta = -1.f;                    // without records: how far the two clouds have moved against each other since the searches that left keys[] (mm, upper bound; < 0: unknown)
  float stretch = 1.f;                   // distances in the searched cloud's canonical frame are at most this times the posed ones (Cloud::pose_stretch, rounded up)
  int dt_max = 12;                       // what dt == 255 stands for (the grid's dt_steps)
  // a BOUNDED query whose ball is wide (scattered among the others: a match that moved far, a long correspondence) is
  // neither walked by its own thread (the wave would wait for it) nor flagged for the culled kernel (one such query
  // per 64 would wake every block): its ordinal is appended here and a second launch gives each of them a whole wave
  uint32_t *wide_list = nullptr, *wide_count = nullptr;
  uint32_t *cull_sets = nullptr, *cull_count = nullptr;      // with heavy, one lane per query: the 64-query sets that hold a flagged query (what launch_nn_cull_list_batch walks)
};
constexpr int kGridBatchPairs = 12;
constexpr int kWideCounters = 64;          // pairs of one fused pass that can have wide lists (more: the pass takes the culled kernel)
constexpr int kGridDtMax = 12;        // most dilation steps of the distance map (a grid is built with as many as the first search radius it serves needs)
constexpr int kGridLightRows = 24;    // default number of rows of cells (x-runs) a thread walks by itself (with the probe at its own threshold of 12 rows: 24 takes 0.19 ms off the first four passes of a window that restarts from the prior, settled passes equal; 32 and 48 within 2 % of it; before the probe had a threshold of its own, on the 12 x 200k ring: 9..16 equal, 4 and 27 slower)
struct GridBatch { GridPair p[kGridBatchPairs]; float cap2; int light_rows; int cluster = 65; int probe = 0; int probe_rows = 1 << 30; float delta_max = 0.f; float cert_margin = 0.f; };
int launch_nn_grid_batch(Ctx *c, const GridPair *pairs, int n_pairs, float cap2, bool fma);
int launch_nn_grid_wide_batch(Ctx *c, const GridPair *pairs, int n_pairs, float cap2, bool fma);
int launch_nn_grid_tail_batch(Ctx *c, const GridPair *pairs, int n_pairs, float cap2, bool fma);     // the two launches below in one
int launch_nn_grid_sets_batch(Ctx *c, const GridPair *pairs, int n_pairs, float cap2, bool fma);     // the 64-query sets the first launch listed (flagged queries only): a block per set over the grid     // the queries the first launch put on the wide lists
// the set's grid (built from `canon`, a cloud holding the set's canonical coordinates) and the posed copy's grid-ordered
// coordinates; false = not available (the caller uses the culled kernel)
int cloud_bbox(Ctx *c, const float4 *pts, size_t n, float out[6]);      // {lo xyz, hi xyz} on the host (synchronises the stream)
bool ensure_grid(Ctx *c, Cloud &canon, double reach);
int ensure_model_grid(Ctx *c, Cloud &t, double reach, bool *ok);      // the grid of a cloud over its own coordinates (Cloud::mgrid)
GridPair make_model_pair(const Cloud &q, const Cloud &t, nnkey_t *keys);
int ensure_grids(Ctx *c, Cloud *const *canon, int count, double reach, hipStream_t on, hipEvent_t after);      // the same for many sets at once, enqueued on `on` (no wait afterwards)      // reach: the search radius the distance map should be able to rule out (mm)
int refresh_grid_coords_batch(Ctx *c, Cloud *const *posed, int count);
int ensure_pose_table(Ctx *c);      // (mvr_ctx.hip) fills the device pose records of the chain being enqueued if nobody has yet
// can the posed cloud's grid-ordered coordinates be written now (grid there and ready, buffer, position maps)?  Queues what that needs
// on the context's stream; the caller then writes gsorted[] and sets gcoords_valid.
int grid_coords_prepare(Ctx *c, Cloud *cl, bool *ok);
int flush_g2h(Ctx *c);      // launches what grid_coords_prepare queued (grid position <-> Hilbert position maps), all grids in one launch      // (*ok = false: no grid to write for; the return value is a status)
// the pipelined pass loop shared by mvr_ring_run and mvr_ring_run_sharded (mvr_ctx.hip).  enqueue(): one pass's GPU work on
// c->stream, ending with the edge table on its way to c->h_table; solve(): the host step on c->h_table, poses in / out.
struct PassLoop {
  int n_views = 0; const int *posed_slots = nullptr, *raw_slots = nullptr; double *poses = nullptr;     // poses: [views][16], in / out
  int (*enqueue)(void *self) = nullptr; int (*solve)(void *self) = nullptr; void *self = nullptr;
  bool ends_with_sums = false;      // enqueue()'s last operation on the stream is the fused sums launch writing the host's table (it may carry the completion word)
  double reach = 0.0;              // the search radius of the passes (mm): with it the loop builds the scans' grids while the first pass searches
  unsigned long long sig = 0;      // identifies the registration (slots, point sets, edges, parameters): a run that ended in steady state lets the next run of the SAME registration start pipelined
};
unsigned long long pass_loop_sig(Ctx *c, int n_views, const int *posed_slots, const int *raw_slots, int ne, const int *edge_src, const int *edge_tgt,
                                 double max_dist, int reciprocal, int fma, int extra);
int ring_passes(Ctx *c, int n_steps, const PassLoop &loop, double timing_ms[3]);
// ---- collectives of a context (mvr_world.cpp); every one of them is a no-op returning MVR_OK without a communicator
enum { kReduceMinI64 = 0, kReduceSumF64 = 1 };
int comm_allreduce(Ctx *c, void *dev_buf, size_t count, int kind);      // in place, on the context's stream
int comm_poll(Ctx *c, bool abort_now = true); // RCCL's asynchronous error state; an error aborts the communicator (abort_now) -> MVR_E_RCCL
bool drain_bounded(Ctx *c, int ms);           // hipStreamSynchronize -- bounded by ms once an abort has failed to free the stream (Ctx::stream_stuck)
int comm_abort(Ctx *c, const char *why);      // ncclCommAbort + release of anything that could hold the stream; returns MVR_E_RCCL
int stream_wait(Ctx *c);                      // hipStreamSynchronize; with a communicator: bounded by wait_timeout_ms and watching comm_poll, a timeout aborts
GridPair make_grid_pair(const Cloud &q, size_t q_begin, size_t q_count, const Cloud &t, nnkey_t *keys);
// ---- a composite target searched part by part (mvr_grid.hip: nn_parts_kernel)
struct PartDesc {
  const float4 *gts; const uint32_t *start; const uint8_t *dt;
  const uint32_t *dir, *recs; int dt_shift, dtdim[3];
  float lo[3], inv_h, h, stretch; int dim[3], dt_max;
  double minv[12];
  uint32_t base, n;
};
// forward search of the queries q.sorted[0, nq) in the `count` parts described at c->d_parts: keys[original query index] =
// (d2 bits, COMPOSITE original target index), kKeyInit = nothing within cap2; heavy[sorted position] = 1 for the queries
// that were NOT answered here (an unbounded ball too wide to walk): the culled kernel takes those
int launch_nn_parts(Ctx *c, const Cloud &q, int count, float cap2, bool fma, nnkey_t *keys, uint8_t *heavy, const uint32_t *qbound = nullptr);
int fill_part_coords(Ctx *c, Cloud &target, GridPart &part);
// keys[original index of sorted[pos]] = by_pos[pos] for the flagged positions (the culled kernel answers flagged queries by sorted position)
int launch_merge_flagged_keys(Ctx *c, const float4 *sorted, const uint8_t *flags, const nnkey_t *by_pos, size_t n, nnkey_t *keys);      // target.gsorted[part.base ...] <- the part's coordinates in its grid's order
int launch_nn_cull(Ctx *c, const Cloud &q, size_t q_begin, size_t q_count, const uint8_t *qflags, const Cloud &t, float cap2,
                   bool fma, nnkey_t *keys);
// one scan pair of a batched global pass (mvr_pair_moments2_batch, culled mode): everything the glue and the
// reduction kernels need, by value (blockIdx.y = pair)
struct GluePair {
  const float4 *src = nullptr, *tgt = nullptr;            // original-order points
  const float4 *qs = nullptr, *ts = nullptr;              // the same points in index (Hilbert) order, w = original index
  const nnkey_t *keys = nullptr, *rkeys = nullptr;        // forward keys [ns] (by original index, or by sorted position: by_pos), reverse keys [nt] (by sorted position)
  const uint32_t *qperm = nullptr, *tinv = nullptr;
  uint8_t *flags = nullptr;                               // [nt], sorted target space
  uint32_t *bound = nullptr;                              // instead of flags: [nt] bits of a matching d2 per target (~0 = not matched)
  uint32_t *list = nullptr, *slot = nullptr;              // [<= nt] flagged sorted positions in order; [nt] position -> list index
  uint32_t *chunks = nullptr, *qcount = nullptr;          // [ceil(nt / 256)] flagged per chunk -> exclusive offsets; number flagged
  unsigned long long nt = 0;
  double *partials = nullptr, *out = nullptr;             // [blocks][29] scratch, 32 doubles result
  unsigned long long q_begin = 0, q_count = 0;
  int blocks = 0, by_pos = 0;
  uint32_t *zero_a = nullptr, *zero_b = nullptr, *zero_c = nullptr;          // optional: two device words the moments launch resets to 0 (the grid search's wide-list counters of this pair)
};
struct GlueBatch {
  GluePair p[kBatchPairs]; double max2; double origin[3]; int reciprocal;
  // optional: the LAST block of the final sums launch to finish stores done_seq into done_word (pinned host memory): the
  // host of a pipelined pass loop learns from the kernel itself that the pass's table is complete, instead of from a
  // stream write-value operation queued behind it (a packet of its own: 4 us + a 9 us gap on the critical path)
  uint32_t *done_word = nullptr, *done_counter = nullptr; uint32_t done_seq = 0;
};
int launch_flag_matched_batch(Ctx *c, const GlueBatch &b, int n_pairs);
int launch_compact_flags_batch(Ctx *c, const GlueBatch &b, int n_pairs);      // flags -> ordered list / slot / count, 3 launches for all pairs
int launch_accept_moments2_batch(Ctx *c, const GlueBatch &b, int n_pairs);
int launch_add_f64(Ctx *c, double *dst, const double *src, size_t n);      // dst[i] += src[i] (one small launch)
int reduce_blocks_for(const Ctx *c, size_t n);

// the same flags compacted (ordered) into list[] with count and the inverse slot[] (sorted position -> list position)
int launch_mark_sorted(Ctx *c, const nnkey_t *keys, const uint32_t *qperm, size_t q_begin, size_t q_count, double max2,
                       const uint32_t *tinv, size_t nt, uint8_t *flags, uint32_t *list, uint32_t *count, uint32_t *slot);
// bound[sorted target position] = bits of the forward d2 of a source that matched it (~0: none): where the reverse search of a matched target starts
// seeds of an align's forward searches (mvr_cull.hip): bound[q] = bits of the distance from query q (sorted position) to target point
// seed[q] (sorted position; >= nt: none -> ~0), with the searches' own formula; and the seeds an align leaves behind: the sorted
// position of every query's match (keys by ORIGINAL index, low word = the match's original index)
int launch_seed_to_bound(Ctx *c, const float4 *qs, size_t nq, const float4 *ts, size_t nt, const uint32_t *seed, bool fma, uint32_t *bound);
int launch_keys_to_seed(Ctx *c, const float4 *qs, size_t nq, const nnkey_t *keys, const uint32_t *tinv, uint32_t *seed);
int launch_seed_bounds(Ctx *c, const nnkey_t *keys, const uint32_t *qperm, size_t q_begin, size_t q_count, double max2,
                       const uint32_t *tinv, size_t nt, uint32_t *bound, uint32_t *seed_out = nullptr, uint32_t *zero_word = nullptr);
// culled-mode reciprocal glue: flag the matched targets (one byte per sorted target position)
int launch_flag_matched(Ctx *c, const nnkey_t *keys, const uint32_t *qperm, size_t q_begin, size_t q_count, double max2,
                        const uint32_t *tinv, size_t nt, uint8_t *flags);

// PointCloud::denoise (mvr_denoise.hip): the cloud is replaced by its kept points, in the reference's output order
int denoise_cloud(Ctx *c, Cloud &cl, int segment_threshold, double triangle_length, size_t *n_kept, size_t *n_components,
                  uint32_t *host_index);
int launch_fill_u64(Ctx *c, nnkey_t *p, size_t n, nnkey_t v);
// keys with LOCAL target indices <-> signed 64-bit keys with GLOBAL target indices (what a MIN all-reduce over
// ranks combines; "no neighbour" = INT64_MAX).  import keeps only the matches this shard owns.
int launch_export_keys(Ctx *c, const nnkey_t *keys, size_t n, const SegTable &st, long long *out);
int launch_import_keys(Ctx *c, const long long *in, size_t n, const SegTable &st, nnkey_t *keys);
int launch_mark(Ctx *c, const nnkey_t *keys, size_t q_begin, size_t q_count, double max2,
                uint32_t *slot, uint32_t *list, uint32_t *count);
// The per-query kernels walk positions [q_begin, q_begin+q_count) and take the
// query index i = qperm ? qperm[pos] : pos.  slot is indexed by tinv ? tinv[j] : j.
// pass 1: match[] + {n, sum p, sum q, sum d2}; then means into moments[0..7]
int launch_pass1(Ctx *c, const float4 *src, const float4 *tgt, const nnkey_t *keys,
                 const nnkey_t *rkeys, const uint32_t *slot, const uint32_t *count, const uint32_t *qperm,
                 const uint32_t *tinv, size_t q_begin, size_t q_count, double max2, bool reciprocal,
                 int32_t *match, double *moments);
// pass 2: sigma = (1/n) sum (q-mean_q)(p-mean_p)^T into moments[8..16]
int launch_pass2(Ctx *c, const float4 *src, const float4 *tgt, const int32_t *match, const uint32_t *qperm,
                 size_t q_begin, size_t q_count, double *moments, const unsigned long long *eval_totals = nullptr,
                 double *host_out = nullptr, uint32_t host_seq = 0);      // host_out: mapped pinned memory that receives moments[0..18] and, at double 64, the word host_seq
// raw second moments about `origin` into out[0..31] (device pointer)
int launch_accept_moments2(Ctx *c, const float4 *src, const float4 *tgt, const nnkey_t *keys, const nnkey_t *rkeys,
                           const uint32_t *slot, const uint32_t *qperm, const uint32_t *tinv, size_t q_begin, size_t q_count,
                           double max2, bool reciprocal, const double origin[3], double *out);
int launch_moments2(Ctx *c, const float4 *src, const float4 *tgt, const int32_t *match, const nnkey_t *keys, const uint32_t *qperm,
                    size_t q_begin, size_t q_count, const double origin[3], double *out);
// K10 (extension): point-to-plane normal equations over the accepted pairs:
// out[0..20] = upper triangle of sum a a^T (a = [p x n, n]), out[21..26] = sum a*d,
// out[27] = count, out[28] = sum d^2  (d = n . (q - p)); device pointer
int launch_p2plane(Ctx *c, const float4 *src, const float4 *tgt, const float4 *tnrm, const int32_t *match,
                   const uint32_t *qperm, size_t q_begin, size_t q_count, double *out);
int launch_rotate_normals_f32(Ctx *c, const float4 *in, float4 *out, size_t n, const float T[16]);
int launch_rotate_normals_f64(Ctx *c, const float4 *in, float4 *out, size_t n, const double T[16]);
int launch_fitness(Ctx *c, const nnkey_t *keys, size_t n, double max_range, double *moments);
int launch_transform_f32(Ctx *c, const float4 *in, float4 *out, size_t n, const float T[16]);
int launch_transform_f64(Ctx *c, const float4 *in, float4 *out, size_t n, const double T[16]);
int launch_unpack_xyz(Ctx *c, const float *packed, float4 *out, size_t n);
int launch_pack_xyz(Ctx *c, const float4 *in, float *packed, size_t n);
int launch_decode_keys(Ctx *c, const nnkey_t *keys, size_t n, uint32_t *idx, float *d2);

// ---- host math (host_math.cpp) ------------------------------------------------
void svd3(const double A[9], double U[9], double S[3], double V[9]);
void umeyama_from_moments(const double mean_src[3], const double mean_tgt[3],
                          const double sigma[9], float T[16], double sv[3]);
// pcl TransformationEstimationPointToPlaneLLS: x = (A^T A)^-1 A^T b, T = Rz(g) Ry(b) Rx(a) + t
int p2plane_solve(const double ata_upper[21], const double atb[6], float T[16]);
int invert6(const double A[36], double Ainv[36]);
int solve_dense(int n, double *A, double *b);

}  // namespace mvr
