// csrc/mvr_internal.h -- shared declarations of libmvr_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "mvr_hip.h"

namespace mvr {

// ---- device data layout ------------------------------------------------------
// A cloud is an array of float4 {x,y,z,1}: the same 16-byte record as
// pcl::PointXYZ (mvr/include/types.h:14), so a host upload is one memcpy and
// every lane moves one point with a single 16-byte access (dwordx4).
struct Cloud {
  float4 *pts = nullptr;
  size_t n = 0;
  size_t cap = 0;
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

// work = `work` unless `h_count` is set: then work = *h_count * per_count (the
// reverse NN pass, whose query count is only known on the device)
struct ProfRec { int family; hipEvent_t a, b; double work; const uint32_t *h_count; double per_count; };

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 0;
  int clock_mhz = 0;
  std::string name;
  std::string last_error;
  Cloud slots[MVR_MAX_SLOTS + 2];          // +2 internal scratch clouds
  // per-pair work buffers (grown on demand)
  nnkey_t *keys = nullptr;   size_t keys_cap = 0;     // [Ns] forward NN keys
  nnkey_t *rkeys = nullptr;  size_t rkeys_cap = 0;    // [Nt'] reverse NN keys (by list position)
  uint32_t *slot = nullptr;  size_t slot_cap = 0;     // [Nt] target -> list position
  uint32_t *list = nullptr;  size_t list_cap = 0;     // [Nt'] distinct matched targets
  int32_t *match = nullptr;  size_t match_cap = 0;    // [Ns] accepted match or -1
  uint32_t *count = nullptr;                          // device counter (list length)
  double *partials = nullptr; size_t partials_cap = 0;
  double *moments = nullptr;                          // device: 64 doubles
  double *h_moments = nullptr;                        // pinned host: 64 doubles
  // launch configuration of the NN kernel (mvr_ctx_tune)
  int nn_q = 8, nn_sub = 32, nn_blocks_per_cu = 2;
  // instrumentation
  bool prof = false;
  std::vector<ProfRec> recs;
  uint32_t *h_counts = nullptr;                       // pinned: device counters copied per profiled launch
  size_t h_counts_used = 0;
  uint64_t prof_launches[MVR_K_COUNT] = {0};
  double prof_ms[MVR_K_COUNT] = {0};
  double prof_work[MVR_K_COUNT] = {0};
};

constexpr int kScratchCur = MVR_MAX_SLOTS;       // ICP's input_transformed
constexpr int kScratchTmp = MVR_MAX_SLOTS + 1;   // fitness temporary

int set_error(Ctx *c, int status, const char *what, hipError_t e = hipSuccess);

#define MVR_HIP_TRY(ctx, expr)                                               \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) return ::mvr::set_error((ctx), MVR_E_HIP, #expr, _e); \
  } while (0)

constexpr size_t kProfCounts = 1 << 16;

struct ProfScope {
  Ctx *c; int fam; hipEvent_t a = nullptr, b = nullptr; double work;
  const uint32_t *d_count = nullptr; double per_count = 0.0;
  ProfScope(Ctx *ctx, int family, double w);
  // work is (*d_count) * per_count, read back asynchronously after the launch
  ProfScope(Ctx *ctx, int family, const uint32_t *dev_count, double per_cnt, double upper_bound);
  ProfScope(const ProfScope &) = delete;
  ProfScope &operator=(const ProfScope &) = delete;
  ~ProfScope();
};

// ---- kernel launchers (mvr_nn.hip / mvr_reduce.hip) -------------------------
// forward / reverse brute-force NN.  Queries: points [q_begin, q_begin+q_count)
// of `q` (direct; key ordinal = point index), or, when `qlist` != null, the
// points q[qlist[k]] for k < *qcount (device counter; q_count is then the upper
// bound used to size the grid; key ordinal = k).
int launch_nn(Ctx *c, const float4 *q, size_t q_begin, size_t q_count, const uint32_t *qlist,
              const uint32_t *qcount, const float4 *t, size_t nt, bool fma, nnkey_t *keys);

int launch_fill_u64(Ctx *c, nnkey_t *p, size_t n, nnkey_t v);
int launch_mark(Ctx *c, const nnkey_t *keys, size_t q_begin, size_t q_count, double max2,
                uint32_t *slot, uint32_t *list, uint32_t *count);
// pass 1: match[] + {n, sum p, sum q, sum d2}; then means into moments[0..7]
int launch_pass1(Ctx *c, const float4 *src, const float4 *tgt, const nnkey_t *keys,
                 const nnkey_t *rkeys, const uint32_t *slot, const uint32_t *count, size_t q_begin,
                 size_t q_count, double max2, bool reciprocal, int32_t *match, double *moments);
// pass 2: sigma = (1/n) sum (q-mean_q)(p-mean_p)^T into moments[8..16]
int launch_pass2(Ctx *c, const float4 *src, const float4 *tgt, const int32_t *match,
                 size_t q_begin, size_t q_count, double *moments);
// raw second moments about `origin` into out[0..31] (device pointer)
int launch_moments2(Ctx *c, const float4 *src, const float4 *tgt, const int32_t *match,
                    size_t q_begin, size_t q_count, const double origin[3], double *out);
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
int invert6(const double A[36], double Ainv[36]);
int solve_dense(int n, double *A, double *b);

}  // namespace mvr
