/*
 * oracle/mvr_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT.
 * PARITY UNPINNED (see mvr_oracle.h): the reference's arithmetic for this path
 * lives in PCL (unpinned, >= 1.7.0; mvr/CMakeLists.txt:10), absent here; the
 * reference holds no tests or golden vectors.  Each function cites the
 * reference call site it restates and the SURVEY.md App. A paragraph that
 * fixes the upstream semantics.
 *
 * Build: gcc -O3 -ffp-contract=off (NO -ffast-math): every float expression
 * below is rounded per operation, in the written order.
 */
#define _GNU_SOURCE
#include "mvr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ params */

API void orc_icp_default_params(orc_icp_params *p)
{
  /* PCL defaults */
  p->use_reciprocal = 0;
  p->max_corr_dist = sqrt(DBL_MAX);
  p->max_iterations = 10;
  p->transformation_epsilon = 0.0;
  p->euclidean_fitness_eps = -DBL_MAX;
  p->fma_dist = 0;
  p->use_kdtree = 1;
}

/* --------------------------------------------------------------- distances */

/* App. A.2: FLANN L2_Simple<float>: result += diff*diff over x,y,z in f32. */
API float orc_dist2(const float *a, const float *b, int fma)
{
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  if (fma)
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

/* -------------------------------------------------------------- transforms */

/* K1 / App. A.1 "input_transformed = transformation * input_transformed":
 * pcl ICP::transformCloud, Eigen Matrix4f * (x,y,z,1). */
API void orc_transform_f32(const float T[16], const float *in, float *out, size_t n)
{
  const float m00 = T[0], m10 = T[1], m20 = T[2];
  const float m01 = T[4], m11 = T[5], m21 = T[6];
  const float m02 = T[8], m12 = T[9], m22 = T[10];
  const float m03 = T[12], m13 = T[13], m23 = T[14];
  for (size_t i = 0; i < n; ++i) {
    float x = in[4 * i], y = in[4 * i + 1], z = in[4 * i + 2];
    float ox = ((m00 * x + m01 * y) + m02 * z) + m03;
    float oy = ((m10 * x + m11 * y) + m12 * z) + m13;
    float oz = ((m20 * x + m21 * y) + m22 * z) + m23;
    out[4 * i] = ox; out[4 * i + 1] = oy; out[4 * i + 2] = oz; out[4 * i + 3] = 1.0f;
  }
}

/* a1: mvr/src/point_cloud.cpp:290-303 -> osg::Matrixd::preMult(const Vec3f&):
 * d = 1/(m[0][3]x + m[1][3]y + m[2][3]z + m[3][3]); each coordinate
 * (m[0][c]x + m[1][c]y + m[2][c]z + m[3][c])*d in double, then to float.
 * OSG's m[r][c] (row-vector) == our column-vector T(c,r) == T[c + 4*r]. */
API void orc_transform_f64(const double T[16], const float *in, float *out, size_t n)
{
  for (size_t i = 0; i < n; ++i) {
    double x = in[4 * i], y = in[4 * i + 1], z = in[4 * i + 2];
    double d = 1.0 / (((T[3] * x + T[7] * y) + T[11] * z) + T[15]);
    double ox = (((T[0] * x + T[4] * y) + T[8] * z) + T[12]) * d;
    double oy = (((T[1] * x + T[5] * y) + T[9] * z) + T[13]) * d;
    double oz = (((T[2] * x + T[6] * y) + T[10] * z) + T[14]) * d;
    out[4 * i] = (float)ox; out[4 * i + 1] = (float)oy; out[4 * i + 2] = (float)oz;
    out[4 * i + 3] = 1.0f;
  }
}

API void orc_mat4f_mul(const float A[16], const float B[16], float C[16])
{
  float R[16];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      float s = A[i] * B[4 * j];
      s = s + A[i + 4] * B[4 * j + 1];
      s = s + A[i + 8] * B[4 * j + 2];
      s = s + A[i + 12] * B[4 * j + 3];
      R[i + 4 * j] = s;
    }
  memcpy(C, R, sizeof R);
}

API void orc_mat4d_mul(const double A[16], const double B[16], double C[16])
{
  double R[16];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      double s = A[i] * B[4 * j];
      s = s + A[i + 4] * B[4 * j + 1];
      s = s + A[i + 8] * B[4 * j + 2];
      s = s + A[i + 12] * B[4 * j + 3];
      R[i + 4 * j] = s;
    }
  memcpy(C, R, sizeof R);
}

/* a2: mvr/src/registrator.cpp:331-342 (translate(-pivot) * rotate(angle,axis)
 * * translate(pivot), OSG row-vector order) as a column-vector matrix. */
API void orc_axis_rotation(const double pivot[3], const double axis[3], double angle,
                           double T[16])
{
  double n = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  double ux = axis[0] / n, uy = axis[1] / n, uz = axis[2] / n;
  double c = cos(angle), s = sin(angle), k = 1.0 - c;
  double R[9] = { c + ux * ux * k,      ux * uy * k - uz * s, ux * uz * k + uy * s,
                  uy * ux * k + uz * s, c + uy * uy * k,      uy * uz * k - ux * s,
                  uz * ux * k - uy * s, uz * uy * k + ux * s, c + uz * uz * k };
  memset(T, 0, 16 * sizeof(double));
  for (int r = 0; r < 3; ++r) {
    for (int cc = 0; cc < 3; ++cc) T[r + 4 * cc] = R[3 * r + cc];
    T[r + 12] = pivot[r] - (R[3 * r] * pivot[0] + R[3 * r + 1] * pivot[1] + R[3 * r + 2] * pivot[2]);
  }
  T[15] = 1.0;
}

/* a2: mvr/src/point_cloud.cpp:409  ((view<7)?(-view):(12-view))*M_PI/6 */
API double orc_turntable_angle(int view, int n_views)
{
  int half = n_views / 2;
  int k = (view <= half) ? -view : (n_views - view);
  return (double)k * (2.0 * M_PI / (double)n_views);
}

/* ------------------------------------------------------------ brute force NN */

API void orc_nn_brute(const float *q, size_t nq, const float *t, size_t nt, int fma,
                      uint32_t *idx, float *d2)
{
  for (size_t i = 0; i < nq; ++i) {
    float best = INFINITY; uint32_t bi = UINT32_MAX;
    const float *qi = q + 4 * i;
    for (size_t j = 0; j < nt; ++j) {
      float d = orc_dist2(qi, t + 4 * j, fma);
      if (d < best) { best = d; bi = (uint32_t)j; }   /* ties keep the lowest index */
    }
    idx[i] = bi; d2[i] = best;
  }
}

/* ------------------------------------------------------------------ kd-tree */
/* The algorithmic class the reference gets from FLANN's KDTreeSingleIndex
 * (leaf size 15 in pcl::KdTreeFLANN).  Exact; ties -> lowest index; pruning is
 * conservative w.r.t. float rounding so results equal orc_nn_brute bit for bit. */

typedef struct { int left, right; int dim; float lo_max, hi_min; uint32_t lo, hi; } kdnode;
typedef struct {
  const float *pts; size_t n; uint32_t *perm; kdnode *nodes; int nnodes, cap;
  float bbmin[3], bbmax[3];
} kdtree;

#define KD_LEAF 15

static int kd_newnode(kdtree *t)
{
  if (t->nnodes == t->cap) { t->cap = t->cap ? 2 * t->cap : 1024; t->nodes = realloc(t->nodes, (size_t)t->cap * sizeof(kdnode)); }
  return t->nnodes++;
}

static void kd_select(const float *pts, uint32_t *a, size_t n, size_t k, int dim)
{
  size_t lo = 0, hi = n - 1;
  while (lo < hi) {
    float pv = pts[4 * (size_t)a[(lo + hi) / 2] + dim];
    size_t i = lo, j = hi;
    for (;;) {
      while (pts[4 * (size_t)a[i] + dim] < pv) ++i;
      while (pts[4 * (size_t)a[j] + dim] > pv) --j;
      if (i >= j) break;
      uint32_t tmp = a[i]; a[i] = a[j]; a[j] = tmp; ++i; --j;
    }
    if (k <= j) hi = j; else lo = j + 1;
  }
}

static int kd_build_rec(kdtree *t, uint32_t lo, uint32_t hi)
{
  int id = kd_newnode(t);
  uint32_t n = hi - lo;
  if (n <= KD_LEAF) {
    t->nodes[id].left = t->nodes[id].right = -1; t->nodes[id].lo = lo; t->nodes[id].hi = hi;
    return id;
  }
  float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (uint32_t k = lo; k < hi; ++k)
    for (int d = 0; d < 3; ++d) {
      float v = t->pts[4 * (size_t)t->perm[k] + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  int dim = 0; float ext = mx[0] - mn[0];
  for (int d = 1; d < 3; ++d) if (mx[d] - mn[d] > ext) { ext = mx[d] - mn[d]; dim = d; }
  uint32_t mid = lo + n / 2;
  kd_select(t->pts, t->perm + lo, n, n / 2, dim);
  float lo_max = -INFINITY, hi_min = INFINITY;
  for (uint32_t k = lo; k < mid; ++k) { float v = t->pts[4 * (size_t)t->perm[k] + dim]; if (v > lo_max) lo_max = v; }
  for (uint32_t k = mid; k < hi; ++k) { float v = t->pts[4 * (size_t)t->perm[k] + dim]; if (v < hi_min) hi_min = v; }
  int l = kd_build_rec(t, lo, mid);
  int r = kd_build_rec(t, mid, hi);
  kdnode *nd = &t->nodes[id];
  nd->left = l; nd->right = r; nd->dim = dim; nd->lo_max = lo_max; nd->hi_min = hi_min;
  nd->lo = lo; nd->hi = hi;
  return id;
}

static kdtree *kd_build(const float *pts, size_t n)
{
  kdtree *t = calloc(1, sizeof *t);
  t->pts = pts; t->n = n;
  t->perm = malloc((n ? n : 1) * sizeof(uint32_t));
  for (size_t i = 0; i < n; ++i) t->perm[i] = (uint32_t)i;
  for (int d = 0; d < 3; ++d) { t->bbmin[d] = INFINITY; t->bbmax[d] = -INFINITY; }
  for (size_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      float v = pts[4 * i + d];
      if (v < t->bbmin[d]) t->bbmin[d] = v;
      if (v > t->bbmax[d]) t->bbmax[d] = v;
    }
  if (n) kd_build_rec(t, 0, (uint32_t)n);
  return t;
}

static void kd_free(kdtree *t) { if (t) { free(t->perm); free(t->nodes); free(t); } }

typedef struct { const kdtree *t; const float *q; int fma; float best; uint32_t bidx; } kdq;

static void kd_search(kdq *s, int id, double mind, double dists[3])
{
  const kdnode *nd = &s->t->nodes[id];
  if (nd->left < 0) {
    for (uint32_t k = nd->lo; k < nd->hi; ++k) {
      uint32_t j = s->t->perm[k];
      float d = orc_dist2(s->q, s->t->pts + 4 * (size_t)j, s->fma);
      if (d < s->best || (d == s->best && j < s->bidx)) { s->best = d; s->bidx = j; }
    }
    return;
  }
  int dim = nd->dim;
  double v = s->q[dim];
  double d1 = v - (double)nd->lo_max, d2 = v - (double)nd->hi_min;
  int nearc, farc; double cut;
  if (d1 + d2 < 0) { nearc = nd->left; farc = nd->right; cut = d2 * d2; }
  else             { nearc = nd->right; farc = nd->left; cut = d1 * d1; }
  /* lo_max <= hi_min, so q is on the near child's side of the far child's
   * closest coordinate and `cut` is a valid lower bound in this dimension;
   * child boxes are nested, hence cut >= the bound already held for `dim`. */
  kd_search(s, nearc, mind, dists);
  double old = dists[dim];
  double nm = mind - old + cut;
  /* explore on <= so that an equal-distance lower index is still found;
   * (1 - 1e-6) keeps the double bound below any float-rounded distance. */
  if (nm * (1.0 - 1e-6) <= (double)s->best) {
    dists[dim] = cut;
    kd_search(s, farc, nm, dists);
    dists[dim] = old;
  }
}

static void kd_nn(const kdtree *t, const float *q, int fma, uint32_t *idx, float *d2)
{
  kdq s = { t, q, fma, INFINITY, UINT32_MAX };
  if (t->n) {
    double dists[3], mind = 0;
    for (int d = 0; d < 3; ++d) {
      double v = q[d], e = 0;
      if (v < t->bbmin[d]) e = (double)t->bbmin[d] - v;
      if (v > t->bbmax[d]) e = v - (double)t->bbmax[d];
      dists[d] = e * e; mind += dists[d];
    }
    kd_search(&s, 0, mind, dists);
  }
  *idx = s.bidx; *d2 = s.best;
}

API void orc_nn_kdtree(const float *q, size_t nq, const float *t, size_t nt, int fma,
                       uint32_t *idx, float *d2)
{
  kdtree *kt = kd_build(t, nt);
  for (size_t i = 0; i < nq; ++i) kd_nn(kt, q + 4 * i, fma, &idx[i], &d2[i]);
  kd_free(kt);
}

/* --------------------------------------------------------- correspondences */

/* a5 / App. A.2.  mvr/src/registrator.cpp:496-502, 644-649 and inside align
 * because of setUseReciprocalCorrespondences(true) (:552, :768, :901). */
API size_t orc_correspondences(const float *src, size_t ns, const float *tgt, size_t nt,
                               double max_dist, int reciprocal, int fma, int use_kdtree,
                               orc_corr *out)
{
  const double max2 = max_dist * max_dist;
  size_t m = 0;
  kdtree *kt = NULL, *ks = NULL;
  if (use_kdtree) { kt = kd_build(tgt, nt); if (reciprocal) ks = kd_build(src, ns); }
  for (size_t i = 0; i < ns; ++i) {
    uint32_t j, i2; float d, dr;
    if (use_kdtree) kd_nn(kt, src + 4 * i, fma, &j, &d);
    else orc_nn_brute(src + 4 * i, 1, tgt, nt, fma, &j, &d);
    if (j == UINT32_MAX) continue;
    if ((double)d > max2) continue;
    if (reciprocal) {
      if (use_kdtree) kd_nn(ks, tgt + 4 * (size_t)j, fma, &i2, &dr);
      else orc_nn_brute(tgt + 4 * (size_t)j, 1, src, ns, fma, &i2, &dr);
      if ((double)dr > max2 || i2 != (uint32_t)i) continue;
    }
    out[m].query = (int32_t)i; out[m].match = (int32_t)j; out[m].dist2 = d; ++m;
  }
  kd_free(kt); kd_free(ks);
  return m;
}

/* The same correspondences with the per-query searches spread over `threads` OpenMP threads -- a courtesy
 * CPU baseline ("OpenMP over queries on all host cores, core count printed"); the reference
 * itself is single-threaded.  Output identical to orc_correspondences (kd-tree), in index order. */
API size_t orc_correspondences_mt(const float *src, size_t ns, const float *tgt, size_t nt, double max_dist,
                                  int reciprocal, int fma, int threads, orc_corr *out)
{
  const double max2 = max_dist * max_dist;
  if (threads < 1) threads = 1;
  kdtree *kt = kd_build(tgt, nt), *ks = reciprocal ? kd_build(src, ns) : NULL;
  int32_t *mj = (int32_t *)malloc((ns ? ns : 1) * sizeof(int32_t));
  float *md = (float *)malloc((ns ? ns : 1) * sizeof(float));
#pragma omp parallel for schedule(dynamic, 2048) num_threads(threads)
  for (long long ii = 0; ii < (long long)ns; ++ii) {
    const size_t i = (size_t)ii;
    uint32_t j, i2; float d, dr;
    mj[i] = -1;
    kd_nn(kt, src + 4 * i, fma, &j, &d);
    if (j == UINT32_MAX || (double)d > max2) continue;
    if (reciprocal) {
      kd_nn(ks, tgt + 4 * (size_t)j, fma, &i2, &dr);
      if ((double)dr > max2 || i2 != (uint32_t)i) continue;
    }
    mj[i] = (int32_t)j; md[i] = d;
  }
  size_t m = 0;
  for (size_t i = 0; i < ns; ++i)
    if (mj[i] >= 0) { out[m].query = (int32_t)i; out[m].match = mj[i]; out[m].dist2 = md[i]; ++m; }
  free(mj); free(md);
  kd_free(kt); kd_free(ks);
  return m;
}

/* --------------------------------------------------------------------- denoise */
static uint32_t uf_find(uint32_t *parent, uint32_t i)
{
  while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; }
  return i;
}

static int cmp_u64(const void *a, const void *b)
{
  const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

API size_t orc_denoise(const float *pts, size_t n, int segment_threshold, double triangle_length,
                       uint32_t *out_index, uint32_t *label, size_t *n_components)
{
  if (n_components) *n_components = 0;
  if (n == 0) return 0;
  uint32_t *parent = (uint32_t *)malloc(n * sizeof(uint32_t));
  for (size_t i = 0; i < n; ++i) parent[i] = (uint32_t)i;
  /* uniform grid with cells >= r: all pairs within r lie in the 27 neighbouring cells */
  double lo[3] = { DBL_MAX, DBL_MAX, DBL_MAX }, hi[3] = { -DBL_MAX, -DBL_MAX, -DBL_MAX };
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) { const double v = pts[4 * i + k]; if (v < lo[k]) lo[k] = v; if (v > hi[k]) hi[k] = v; }
  double h = triangle_length > 0 ? triangle_length : 1e-30;
  for (int k = 0; k < 3; ++k) if ((hi[k] - lo[k]) / h > 1000000.0) h = (hi[k] - lo[k]) / 1000000.0;
  /* sort (cell key, index) */
  uint64_t *keyidx = (uint64_t *)malloc(n * 2 * sizeof(uint64_t));
  for (size_t i = 0; i < n; ++i) {
    uint64_t c[3];
    for (int k = 0; k < 3; ++k) c[k] = (uint64_t)((pts[4 * i + k] - lo[k]) / h);
    keyidx[2 * i] = (c[2] << 42) | (c[1] << 21) | c[0];
    keyidx[2 * i + 1] = i;
  }
  qsort(keyidx, n, 2 * sizeof(uint64_t), cmp_u64);
  for (size_t a = 0; a < n; ++a) {
    const uint64_t ka = keyidx[2 * a];
    const size_t i = (size_t)keyidx[2 * a + 1];
    const int64_t cx = (int64_t)(ka & 0x1FFFFF), cy = (int64_t)((ka >> 21) & 0x1FFFFF), cz = (int64_t)(ka >> 42);
    for (int64_t dz = -1; dz <= 1; ++dz)
      for (int64_t dy = -1; dy <= 1; ++dy) {
        if (cz + dz < 0 || cy + dy < 0) continue;
        const int64_t x0 = cx > 0 ? cx - 1 : 0;
        const uint64_t k0 = ((uint64_t)(cz + dz) << 42) | ((uint64_t)(cy + dy) << 21) | (uint64_t)x0;
        const uint64_t k1 = ((uint64_t)(cz + dz) << 42) | ((uint64_t)(cy + dy) << 21) | (uint64_t)(cx + 1);
        /* lower bound of k0 */
        size_t lo_i = 0, hi_i = n;
        while (lo_i < hi_i) { const size_t mid = (lo_i + hi_i) / 2; if (keyidx[2 * mid] < k0) lo_i = mid + 1; else hi_i = mid; }
        for (size_t b = lo_i; b < n && keyidx[2 * b] <= k1; ++b) {
          const size_t j = (size_t)keyidx[2 * b + 1];
          if (j >= i) continue;
          const double dx = (double)pts[4 * i] - (double)pts[4 * j], dyy = (double)pts[4 * i + 1] - (double)pts[4 * j + 1],
                       dzz = (double)pts[4 * i + 2] - (double)pts[4 * j + 2];
          if (sqrt(dx * dx + dyy * dyy + dzz * dzz) > triangle_length) continue;   /* point_cloud.cpp:490-491 */
          uint32_t ra = uf_find(parent, (uint32_t)i), rb = uf_find(parent, (uint32_t)j);
          if (ra != rb) { if (ra < rb) parent[rb] = ra; else parent[ra] = rb; }      /* root = smallest index */
        }
      }
  }
  uint32_t *size = (uint32_t *)calloc(n, sizeof(uint32_t));
  for (size_t i = 0; i < n; ++i) { const uint32_t r = uf_find(parent, (uint32_t)i); parent[i] = r; ++size[r]; if (label) label[i] = r; }
  size_t ncomp = 0, kept = 0;
  /* components in the order of their smallest index (boost::connected_components numbering), members ascending */
  uint64_t *order = (uint64_t *)malloc(n * sizeof(uint64_t));
  for (size_t i = 0; i < n; ++i) { if (parent[i] == i) ++ncomp; order[i] = ((uint64_t)parent[i] << 32) | (uint64_t)i; }
  qsort(order, n, sizeof(uint64_t), cmp_u64);
  for (size_t k = 0; k < n; ++k) {
    const uint32_t i = (uint32_t)order[k], r = (uint32_t)(order[k] >> 32);
    if ((int64_t)size[r] >= (int64_t)segment_threshold) out_index[kept++] = i;
  }
  if (n_components) *n_components = ncomp;
  free(order); free(size); free(keyidx); free(parent);
  return kept;
}

/* ------------------------------------------------------------------- SVD 3x3 */

static double det3(const double M[9])
{
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6])
       + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* One-sided (Hestenes) Jacobi SVD, row-major 3x3, singular values sorted
 * descending as Eigen::JacobiSVD does. */
API void orc_svd3(const double A[9], double U[9], double S[3], double V[9])
{
  double B[9], W[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
  memcpy(B, A, sizeof B);
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int r = 0; r < 3; ++r) { al += B[3 * r + p] * B[3 * r + p]; be += B[3 * r + q] * B[3 * r + q]; ga += B[3 * r + p] * B[3 * r + q]; }
        if (ga == 0.0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
        rotated = 1;
        double zeta = (be - al) / (2.0 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          double bp = B[3 * r + p], bq = B[3 * r + q];
          B[3 * r + p] = c * bp - s * bq; B[3 * r + q] = s * bp + c * bq;
          double wp = W[3 * r + p], wq = W[3 * r + q];
          W[3 * r + p] = c * wp - s * wq; W[3 * r + q] = s * wp + c * wq;
        }
      }
    if (!rotated) break;
  }
  double sv[3]; int ord[3] = { 0, 1, 2 };
  for (int j = 0; j < 3; ++j) sv[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
  for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (sv[ord[b]] > sv[ord[a]]) { int tmp = ord[a]; ord[a] = ord[b]; ord[b] = tmp; }
  for (int j = 0; j < 3; ++j) {
    int o = ord[j]; S[j] = sv[o];
    for (int r = 0; r < 3; ++r) { V[3 * r + j] = W[3 * r + o]; U[3 * r + j] = (sv[o] > 0) ? B[3 * r + o] / sv[o] : 0.0; }
  }
  /* complete U for (numerically) zero singular values */
  double tiny = S[0] * 1e-300 + DBL_MIN;
  if (S[0] <= tiny) { double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }; memcpy(U, I, sizeof I); return; }
  if (S[1] <= S[0] * 1e-15) {
    /* rank 1: pick any unit vector orthogonal to u0 */
    double u0[3] = { U[0], U[3], U[6] };
    int k = (fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2])) ? 0 : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
    double e[3] = { 0, 0, 0 }; e[k] = 1;
    double d = u0[k];
    double u1[3] = { e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2] };
    double n = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    U[1] = u1[0] / n; U[4] = u1[1] / n; U[7] = u1[2] / n;
  }
  if (S[2] <= S[0] * 1e-15) {
    double a[3] = { U[0], U[3], U[6] }, b[3] = { U[1], U[4], U[7] };
    U[2] = a[1] * b[2] - a[2] * b[1]; U[5] = a[2] * b[0] - a[0] * b[2]; U[8] = a[0] * b[1] - a[1] * b[0];
  }
}

/* a6 / App. A.3: Eigen::umeyama(src, dst, with_scaling=false), Eigen 3.3 sign rule. */
API void orc_umeyama_from_moments(const double mean_src[3], const double mean_tgt[3],
                                  const double sigma[9], float T[16], double svout[3])
{
  double U[9], S[3], V[9], R[9];
  orc_svd3(sigma, U, S, V);
  double sgn = (det3(U) * det3(V) < 0) ? -1.0 : 1.0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      R[3 * r + c] = U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1] + sgn * U[3 * r + 2] * V[3 * c + 2];
  memset(T, 0, 16 * sizeof(float));
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[r + 4 * c] = (float)R[3 * r + c];
    double t = mean_tgt[r] - (R[3 * r] * mean_src[0] + R[3 * r + 1] * mean_src[1] + R[3 * r + 2] * mean_src[2]);
    T[r + 12] = (float)t;
  }
  T[15] = 1.0f;
  if (svout) { svout[0] = S[0]; svout[1] = S[1]; svout[2] = S[2]; }
}

API int orc_umeyama(const float *src, const float *tgt, const orc_corr *c, size_t m,
                    float T[16], double *mom)
{
  if (m < 3) return -1;
  double ms[3] = { 0, 0, 0 }, mt[3] = { 0, 0, 0 }, sd = 0;
  for (size_t k = 0; k < m; ++k) {
    const float *p = src + 4 * (size_t)c[k].query, *q = tgt + 4 * (size_t)c[k].match;
    for (int d = 0; d < 3; ++d) { ms[d] += p[d]; mt[d] += q[d]; }
    sd += c[k].dist2;
  }
  for (int d = 0; d < 3; ++d) { ms[d] /= (double)m; mt[d] /= (double)m; }
  double sig[9] = { 0 };
  for (size_t k = 0; k < m; ++k) {
    const float *p = src + 4 * (size_t)c[k].query, *q = tgt + 4 * (size_t)c[k].match;
    double dp[3] = { p[0] - ms[0], p[1] - ms[1], p[2] - ms[2] };
    double dq[3] = { q[0] - mt[0], q[1] - mt[1], q[2] - mt[2] };
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) sig[3 * r + cc] += dq[r] * dp[cc];
  }
  for (int k = 0; k < 9; ++k) sig[k] /= (double)m;
  double sv[3];
  orc_umeyama_from_moments(ms, mt, sig, T, sv);
  if (mom) {
    mom[0] = (double)m;
    for (int d = 0; d < 3; ++d) { mom[1 + d] = ms[d]; mom[4 + d] = mt[d]; }
    mom[7] = sd / (double)m;
    for (int k = 0; k < 9; ++k) mom[8 + k] = sig[k];
    mom[17] = sv[0]; mom[18] = sv[1]; mom[19] = sv[2];
  }
  return 0;
}

/* Eigen's ACTUAL arithmetic for the same estimate (Scalar = float throughout): a model of
 * Eigen::umeyama as TransformationEstimationSVD calls it, used only to MEASURE how far the f64-accumulating oracle
 * above sits from the f32 reference arithmetic (tests/test_oracle.py).
 *   means : src.rowwise().sum() * (1/M) -- a row of a column-major 3xM matrix is strided, so Eigen's reduction is
 *           the plain sequential one: ((p0 + p1) + p2) + ... in float;
 *   sigma : (1/M) * dst_demean * src_demean^T in float; Eigen's product kernel accumulates over blocks of the inner
 *           dimension: `block` = 0 models one sequential pass, > 0 sums partial products of `block` columns each
 *           and adds the block sums in order (the kc blocking of the GEBP kernel);
 *   SVD   : JacobiSVD<Matrix3f>; modelled by the f64 SVD of the float sigma (its own error, ~1e-7, is far below
 *           the effects measured here), R rounded to float, t = mean_dst - R * mean_src in float. */
API int orc_umeyama_f32(const float *src, const float *tgt, const orc_corr *c, size_t m, int block, float T[16])
{
  if (m < 3) return -1;
  const float inv = 1.0f / (float)m;
  float ms[3] = { 0, 0, 0 }, mt[3] = { 0, 0, 0 };
  for (size_t k = 0; k < m; ++k) {
    const float *p = src + 4 * (size_t)c[k].query, *q = tgt + 4 * (size_t)c[k].match;
    for (int d = 0; d < 3; ++d) { ms[d] = ms[d] + p[d]; mt[d] = mt[d] + q[d]; }
  }
  for (int d = 0; d < 3; ++d) { ms[d] = ms[d] * inv; mt[d] = mt[d] * inv; }
  float sig[9] = { 0 }, part[9] = { 0 };
  size_t in_block = 0;
  for (size_t k = 0; k < m; ++k) {
    const float *p = src + 4 * (size_t)c[k].query, *q = tgt + 4 * (size_t)c[k].match;
    const float dp[3] = { p[0] - ms[0], p[1] - ms[1], p[2] - ms[2] };
    const float dq[3] = { q[0] - mt[0], q[1] - mt[1], q[2] - mt[2] };
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) { const float pr = dq[r] * dp[cc]; part[3 * r + cc] = part[3 * r + cc] + pr; }
    if (block > 0 && ++in_block == (size_t)block) {
      for (int j = 0; j < 9; ++j) { sig[j] = sig[j] + part[j]; part[j] = 0; }
      in_block = 0;
    }
  }
  for (int j = 0; j < 9; ++j) sig[j] = (sig[j] + part[j]) * inv;
  double sd[9], U[9], S[3], V[9];
  for (int j = 0; j < 9; ++j) sd[j] = sig[j];
  orc_svd3(sd, U, S, V);
  const double sgn = (det3(U) * det3(V) < 0) ? -1.0 : 1.0;
  float R[9];
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc)
      R[3 * r + cc] = (float)(U[3 * r] * V[3 * cc] + U[3 * r + 1] * V[3 * cc + 1] + sgn * U[3 * r + 2] * V[3 * cc + 2]);
  memset(T, 0, 16 * sizeof(float));
  for (int r = 0; r < 3; ++r) {
    for (int cc = 0; cc < 3; ++cc) T[r + 4 * cc] = R[3 * r + cc];
    float rp = R[3 * r] * ms[0];
    rp = rp + R[3 * r + 1] * ms[1];
    rp = rp + R[3 * r + 2] * ms[2];
    T[r + 12] = mt[r] - rp;
  }
  T[15] = 1.0f;
  return 0;
}

/* --------------------------------------------------------------------- ICP */

/* a4 + a7 / App. A.1 + A.4.  Call sites mvr/src/registrator.cpp:569,920,1012,1024. */
/* EXTENSION: pcl TransformationEstimationPointToPlaneLLS (all sums in double, in
 * correspondence order).  sums29: 21 upper-triangle of A^T A, 6 of A^T d, count, sum d^2. */
API int orc_p2plane(const float *src, const float *tgt, const float *tnrm, const orc_corr *c, size_t m,
                    float T[16], double *sums29)
{
  if (m < 3) return -1;
  double acc[29] = { 0 };
  for (size_t k = 0; k < m; ++k) {
    const float *p = src + 4 * (size_t)c[k].query, *q = tgt + 4 * (size_t)c[k].match, *n = tnrm + 4 * (size_t)c[k].match;
    double sx = p[0], sy = p[1], sz = p[2], nx = n[0], ny = n[1], nz = n[2];
    double a[6] = { nz * sy - ny * sz, nx * sz - nz * sx, ny * sx - nx * sy, nx, ny, nz };
    double d = nx * (double)q[0] + ny * (double)q[1] + nz * (double)q[2] - nx * sx - ny * sy - nz * sz;
    int t = 0;
    for (int r = 0; r < 6; ++r) for (int cc = r; cc < 6; ++cc) acc[t++] += a[r] * a[cc];
    for (int r = 0; r < 6; ++r) acc[21 + r] += a[r] * d;
    acc[27] += 1.0; acc[28] += d * d;
  }
  if (sums29) memcpy(sums29, acc, sizeof acc);
  double A[36], Ainv[36], x[6] = { 0 };
  int t = 0;
  for (int r = 0; r < 6; ++r) for (int cc = r; cc < 6; ++cc) { A[6 * r + cc] = A[6 * cc + r] = acc[t++]; }
  if (orc_invert6(A, Ainv) != 0) return -2;
  for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) x[r] += Ainv[6 * r + cc] * acc[21 + cc];
  /* constructTransformationMatrix(alpha, beta, gamma, tx, ty, tz) */
  double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
  memset(T, 0, 16 * sizeof(float));
  T[0] = (float)(cg * cb); T[4] = (float)(-sg * ca + cg * sb * sa); T[8]  = (float)(sg * sa + cg * sb * ca);  T[12] = (float)x[3];
  T[1] = (float)(sg * cb); T[5] = (float)(cg * ca + sg * sb * sa);  T[9]  = (float)(-cg * sa + sg * sb * ca); T[13] = (float)x[4];
  T[2] = (float)(-sb);     T[6] = (float)(cb * sa);                 T[10] = (float)(cb * ca);                 T[14] = (float)x[5];
  T[15] = 1.0f;
  return 0;
}

static int icp_align_impl(const float *src, size_t ns, const float *tgt, const float *tnrm, size_t nt,
                          const orc_icp_params *p, float *out, float T[16], orc_icp_stats *st)
{
  float *cur = malloc((ns ? ns : 1) * 16);
  orc_corr *corr = malloc((ns ? ns : 1) * sizeof(orc_corr));
  for (size_t i = 0; i < ns; ++i) { cur[4 * i] = src[4 * i]; cur[4 * i + 1] = src[4 * i + 1]; cur[4 * i + 2] = src[4 * i + 2]; cur[4 * i + 3] = 1.0f; }
  float fin[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
  float tr[16];
  memcpy(tr, fin, sizeof tr);
  /* DefaultConvergenceCriteria state */
  const double rot_thr = 1.0 - p->transformation_epsilon, trans_thr = p->transformation_epsilon;
  const double rel_mse = p->euclidean_fitness_eps, abs_mse = 1e-12;
  double prev_mse = DBL_MAX, cur_mse = 0;
  int iters = 0, converged = 0, state = ORC_CONV_NOT, ncorr = 0;
  double evals = 0;
  do {
    size_t m = orc_correspondences(cur, ns, tgt, nt, p->max_corr_dist, p->use_reciprocal,
                                   p->fma_dist, p->use_kdtree, corr);
    evals += (double)ns * (double)nt;
    ncorr = (int)m;
    if (m < 3) { state = ORC_CONV_NO_CORRESPONDENCES; converged = 0; break; }
    double mom[20];
    orc_umeyama(cur, tgt, corr, m, tr, mom);
    cur_mse = mom[7];
    if (tnrm && orc_p2plane(cur, tgt, tnrm, corr, m, tr, NULL) != 0) { state = ORC_CONV_NO_CORRESPONDENCES; converged = 0; break; }
    orc_transform_f32(tr, cur, cur, ns);
    orc_mat4f_mul(tr, fin, fin);
    ++iters;
    /* hasConverged() */
    state = ORC_CONV_NOT; converged = 0;
    if (iters >= p->max_iterations) { state = ORC_CONV_ITERATIONS; converged = 1; }
    else {
      double cos_angle = 0.5 * ((double)tr[0] + (double)tr[5] + (double)tr[10] - 1.0);
      double t2 = (double)tr[12] * (double)tr[12] + (double)tr[13] * (double)tr[13] + (double)tr[14] * (double)tr[14];
      if (cos_angle >= rot_thr && t2 <= trans_thr) { state = ORC_CONV_TRANSFORM; converged = 1; }
      else if (fabs(cur_mse - prev_mse) < abs_mse) { state = ORC_CONV_ABS_MSE; converged = 1; }
      else if (fabs(cur_mse - prev_mse) / prev_mse < rel_mse) { state = ORC_CONV_REL_MSE; converged = 1; }
      else prev_mse = cur_mse;
    }
  } while (!converged);
  /* output = final * (*input), recomputed from the ORIGINAL input (alias-safe) */
  if (out) orc_transform_f32(fin, src, out, ns);
  memcpy(T, fin, 16 * sizeof(float));
  if (st) { st->iterations = iters; st->converged = converged; st->state = state; st->n_corr = ncorr; st->mse = cur_mse; st->evals = evals; }
  free(cur); free(corr);
  return state == ORC_CONV_NO_CORRESPONDENCES ? -1 : 0;
}

API int orc_icp_align(const float *src, size_t ns, const float *tgt, size_t nt,
                      const orc_icp_params *p, float *out, float T[16], orc_icp_stats *st)
{
  return icp_align_impl(src, ns, tgt, NULL, nt, p, out, T, st);
}

API int orc_icp_align_p2plane(const float *src, size_t ns, const float *tgt, const float *tnrm, size_t nt,
                              const orc_icp_params *p, float *out, float T[16], orc_icp_stats *st)
{
  return icp_align_impl(src, ns, tgt, tnrm, nt, p, out, T, st);
}

/* a8 / App. A.5.  mvr/src/registrator.cpp:572,923,1015. */
API double orc_fitness(const float *input, size_t ns, const float *tgt, size_t nt,
                       const float T[16], double max_range, int fma, int use_kdtree)
{
  float *tmp = malloc((ns ? ns : 1) * 16);
  uint32_t *idx = malloc((ns ? ns : 1) * sizeof(uint32_t));
  float *d2 = malloc((ns ? ns : 1) * sizeof(float));
  orc_transform_f32(T, input, tmp, ns);
  if (use_kdtree) orc_nn_kdtree(tmp, ns, tgt, nt, fma, idx, d2);
  else orc_nn_brute(tmp, ns, tgt, nt, fma, idx, d2);
  double sum = 0; size_t nr = 0;
  for (size_t i = 0; i < ns; ++i)
    if (idx[i] != UINT32_MAX && (double)d2[i] <= max_range) { sum += d2[i]; ++nr; }
  free(tmp); free(idx); free(d2);
  return nr ? sum / (double)nr : DBL_MAX;
}

/* --------------------------------------------------------------------- LUM */

/* pcl::getTransformation(x,y,z,roll,pitch,yaw) = T * Rz(yaw) Ry(pitch) Rx(roll) */
API void orc_pose_to_mat4(const double pose[6], double T[16])
{
  double A = cos(pose[5]), B = sin(pose[5]), C = cos(pose[4]), D = sin(pose[4]);
  double E = cos(pose[3]), F = sin(pose[3]), DE = D * E, DF = D * F;
  memset(T, 0, 16 * sizeof(double));
  T[0] = A * C;  T[4] = A * DF - B * E;  T[8]  = B * F + A * DE;  T[12] = pose[0];
  T[1] = B * C;  T[5] = A * E + B * DF;  T[9]  = B * DE - A * F;  T[13] = pose[1];
  T[2] = -D;     T[6] = C * F;           T[10] = C * E;           T[14] = pose[2];
  T[15] = 1.0;
}

API int orc_solve_dense(int n, double *A, double *b)
{
  for (int k = 0; k < n; ++k) {
    int piv = k; double mx = fabs(A[k * n + k]);
    for (int r = k + 1; r < n; ++r) if (fabs(A[r * n + k]) > mx) { mx = fabs(A[r * n + k]); piv = r; }
    if (mx == 0.0) return -1;
    if (piv != k) {
      for (int c = 0; c < n; ++c) { double t = A[k * n + c]; A[k * n + c] = A[piv * n + c]; A[piv * n + c] = t; }
      double t = b[k]; b[k] = b[piv]; b[piv] = t;
    }
    for (int r = k + 1; r < n; ++r) {
      double f = A[r * n + k] / A[k * n + k];
      if (f == 0.0) continue;
      for (int c = k; c < n; ++c) A[r * n + c] -= f * A[k * n + c];
      b[r] -= f * b[k];
    }
  }
  for (int k = n - 1; k >= 0; --k) {
    double s = b[k];
    for (int c = k + 1; c < n; ++c) s -= A[k * n + c] * b[c];
    b[k] = s / A[k * n + k];
  }
  return 0;
}

API int orc_invert6(const double A[36], double Ainv[36])
{
  double M[6][12];
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { M[r][c] = A[6 * r + c]; M[r][6 + c] = (r == c); }
  for (int k = 0; k < 6; ++k) {
    int piv = k; double mx = fabs(M[k][k]);
    for (int r = k + 1; r < 6; ++r) if (fabs(M[r][k]) > mx) { mx = fabs(M[r][k]); piv = r; }
    if (mx == 0.0) return -1;
    if (piv != k) for (int c = 0; c < 12; ++c) { double t = M[k][c]; M[k][c] = M[piv][c]; M[piv][c] = t; }
    double inv = 1.0 / M[k][k];
    for (int c = 0; c < 12; ++c) M[k][c] *= inv;
    for (int r = 0; r < 6; ++r) if (r != k) { double f = M[r][k]; if (f != 0.0) for (int c = 0; c < 12; ++c) M[r][c] -= f * M[k][c]; }
  }
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Ainv[6 * r + c] = M[r][6 + c];
  return 0;
}

/* LUM::computeEdge (App. A.6): sums over correspondences in index order. */
API size_t orc_lum_edge(const float *src, const float *tgt, const orc_corr *c, size_t m,
                        const double pose_s[6], const double pose_t[6],
                        double MM[36], double MZ[6], double *ss)
{
  double Ts[16], Tt[16];
  orc_pose_to_mat4(pose_s, Ts); orc_pose_to_mat4(pose_t, Tt);
  memset(MM, 0, 36 * sizeof(double)); memset(MZ, 0, 6 * sizeof(double)); *ss = 0;
  double *av = malloc((m ? m : 1) * 3 * sizeof(double)), *df = malloc((m ? m : 1) * 3 * sizeof(double));
  size_t oci = 0;
  for (size_t k = 0; k < m; ++k) {
    const float *p = src + 4 * (size_t)c[k].query, *q = tgt + 4 * (size_t)c[k].match;
    double a[3], b[3];
    for (int r = 0; r < 3; ++r) {
      a[r] = Ts[r] * p[0] + Ts[r + 4] * p[1] + Ts[r + 8] * p[2] + Ts[r + 12];
      b[r] = Tt[r] * q[0] + Tt[r + 4] * q[1] + Tt[r + 8] * q[2] + Tt[r + 12];
    }
    if (!isfinite(a[0]) || !isfinite(a[1]) || !isfinite(a[2]) || !isfinite(b[0]) || !isfinite(b[1]) || !isfinite(b[2])) continue;
    for (int r = 0; r < 3; ++r) { av[3 * oci + r] = 0.5 * (a[r] + b[r]); df[3 * oci + r] = a[r] - b[r]; }
    ++oci;
  }
  if (oci < 3) { free(av); free(df); return oci; }
#define MMr(r, c) MM[6 * (r) + (c)]
  for (size_t k = 0; k < oci; ++k) {
    double x = av[3 * k], y = av[3 * k + 1], z = av[3 * k + 2];
    double dx = df[3 * k], dy = df[3 * k + 1], dz = df[3 * k + 2];
    MMr(0, 4) -= y; MMr(0, 5) += z; MMr(1, 3) -= z; MMr(1, 4) += x; MMr(2, 3) += y; MMr(2, 5) -= x;
    MMr(3, 4) -= x * z; MMr(3, 5) -= x * y; MMr(4, 5) -= y * z;
    MMr(3, 3) += y * y + z * z; MMr(4, 4) += x * x + y * y; MMr(5, 5) += x * x + z * z;
    MZ[0] += dx; MZ[1] += dy; MZ[2] += dz;
    MZ[3] += y * dz - z * dy; MZ[4] += x * dy - y * dx; MZ[5] += z * dx - x * dz;
  }
  MMr(0, 0) = MMr(1, 1) = MMr(2, 2) = (double)oci;
  MMr(4, 0) = MMr(0, 4); MMr(5, 0) = MMr(0, 5); MMr(3, 1) = MMr(1, 3); MMr(4, 1) = MMr(1, 4);
  MMr(3, 2) = MMr(2, 3); MMr(5, 2) = MMr(2, 5); MMr(4, 3) = MMr(3, 4); MMr(5, 3) = MMr(3, 5);
  MMr(5, 4) = MMr(4, 5);
#undef MMr
  double Minv[36], D[6] = { 0 };
  if (orc_invert6(MM, Minv) == 0)
    for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) D[r] += Minv[6 * r + cc] * MZ[cc];
  else { *ss = NAN; free(av); free(df); return oci; }
  double s = 0;
  for (size_t k = 0; k < oci; ++k) {
    double x = av[3 * k], y = av[3 * k + 1], z = av[3 * k + 2];
    double e0 = df[3 * k]     - (D[0] + z * D[5] - y * D[4]);
    double e1 = df[3 * k + 1] - (D[1] + x * D[4] - z * D[3]);
    double e2 = df[3 * k + 2] - (D[2] + y * D[3] - x * D[5]);
    s += e0 * e0 + e1 * e1 + e2 * e2;
  }
  *ss = s;
  free(av); free(df);
  return oci;
}

/* LUM::incidenceCorrection (App. A.6): H(X) with  d(R(theta) p + t)/dX = M(p') H(X),  p' = R p + t, for Borrmann et
 * al.'s rotation R = Rx(tx) Ry(ty) Rz(tz) and the M of computeEdge above, M = [I | ex x p', ez x p', ey x p'] (the
 * rotational unknowns are ordered x, z, y).  Rows 3..5 are the rotation vector (x, z, y components) of the three angle
 * increments, w = dtx ex + dty Rx ey + dtz Rx Ry ez; column 5 of the top block is t x (Rx Ry ez).
 * SURVEY App. A.6 recalled that column with sin/cos of the pitch swapped and marked the pattern "least certain";
 * the form below is the one for which the identity holds exactly at any pose (tests/test_oracle.py pins it with a
 * numeric Jacobian at z ~ 917 mm and non-zero poses). */
API void orc_lum_incidence(const double pose[6], double out[36])
{
  memset(out, 0, 36 * sizeof(double));
  for (int k = 0; k < 6; ++k) out[7 * k] = 1.0;
  double cx = cos(pose[3]), sx = sin(pose[3]), cy = cos(pose[4]), sy = sin(pose[4]);
  out[6 * 0 + 4] = pose[1] * sx - pose[2] * cx;
  out[6 * 0 + 5] = pose[1] * cx * cy + pose[2] * sx * cy;
  out[6 * 1 + 3] = pose[2];
  out[6 * 1 + 4] = -pose[0] * sx;
  out[6 * 1 + 5] = -pose[0] * cx * cy + pose[2] * sy;
  out[6 * 2 + 3] = -pose[1];
  out[6 * 2 + 4] = pose[0] * cx;
  out[6 * 2 + 5] = -pose[0] * sx * cy - pose[1] * sy;
  out[6 * 3 + 5] = sy;
  out[6 * 4 + 4] = sx;
  out[6 * 4 + 5] = cx * cy;
  out[6 * 5 + 4] = cx;
  out[6 * 5 + 5] = -sx * cy;
}
#define lum_incidence orc_lum_incidence

/* LUM::compute (App. A.6); driver call site mvr/src/registrator.cpp:653-654. */
API int orc_lum_compute(int n, const float *const *clouds, int ne, const int *es,
                        const int *et, const orc_corr *const *corr, const size_t *ncorr,
                        int max_iterations, double convergence_threshold, double *poses)
{
  if (n < 2) return 0;
  int dim = 6 * (n - 1), it;
  double *G = malloc((size_t)dim * dim * sizeof(double)), *B = malloc((size_t)dim * sizeof(double));
  double *cinv = malloc((size_t)ne * 36 * sizeof(double)), *cinvd = malloc((size_t)ne * 6 * sizeof(double));
  for (it = 0; it < max_iterations; ++it) {
    for (int e = 0; e < ne; ++e) {
      double MM[36], MZ[6], ss;
      size_t oci = orc_lum_edge(clouds[es[e]], clouds[et[e]], corr[e], ncorr[e],
                                poses + 6 * es[e], poses + 6 * et[e], MM, MZ, &ss);
      if (oci < 3 || ss < 0.0000000000001 || !isfinite(ss)) {
        memset(cinv + 36 * e, 0, 36 * sizeof(double)); memset(cinvd + 6 * e, 0, 6 * sizeof(double));
      } else {
        for (int k = 0; k < 36; ++k) cinv[36 * e + k] = MM[k] * (1.0 / ss);
        for (int k = 0; k < 6; ++k) cinvd[6 * e + k] = MZ[k] * (1.0 / ss);
      }
    }
    memset(G, 0, (size_t)dim * dim * sizeof(double)); memset(B, 0, (size_t)dim * sizeof(double));
    for (int vi = 1; vi < n; ++vi)
      for (int vj = 0; vj < n; ++vj) {
        int e = -1, fwd = 0;
        for (int k = 0; k < ne; ++k) if (es[k] == vi && et[k] == vj) { e = k; fwd = 1; break; }
        if (e < 0) for (int k = 0; k < ne; ++k) if (es[k] == vj && et[k] == vi) { e = k; break; }
        if (e < 0) continue;
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) {
            if (vj > 0) G[(size_t)(6 * (vi - 1) + r) * dim + 6 * (vj - 1) + c] = -cinv[36 * e + 6 * r + c];
            G[(size_t)(6 * (vi - 1) + r) * dim + 6 * (vi - 1) + c] += cinv[36 * e + 6 * r + c];
          }
        for (int r = 0; r < 6; ++r) B[6 * (vi - 1) + r] += (fwd ? 1.0 : -1.0) * cinvd[6 * e + r];
      }
    if (orc_solve_dense(dim, G, B) != 0) break;
    double sum = 0;
    for (int vi = 1; vi < n; ++vi) {
      double inc[36], incinv[36], dp[6];
      lum_incidence(poses + 6 * vi, inc);
      if (orc_invert6(inc, incinv) != 0) continue;
      double nrm = 0;
      for (int r = 0; r < 6; ++r) {
        double s = 0;
        for (int c = 0; c < 6; ++c) s += incinv[6 * r + c] * B[6 * (vi - 1) + c];
        dp[r] = -s; nrm += dp[r] * dp[r];
      }
      sum += sqrt(nrm);
      for (int r = 0; r < 6; ++r) poses[6 * vi + r] += dp[r];
    }
    if (sum <= convergence_threshold * (double)(n - 1)) { ++it; break; }
  }
  free(G); free(B); free(cinv); free(cinvd);
  return it;
}

/* ------------------------------------------------------------ refineAxis */

/* min |A x - b| for a full-rank rows x 3 system by Householder QR -- what LAPACK dgels does for the reference
 * (math_solvers::least_squares, mvr/src/math_solvers.cpp:12-38).  A is rows x 3 row-major; both are overwritten. */
static int lstsq3(double *A, double *b, int rows, double x[3])
{
  for (int k = 0; k < 3; ++k) {
    double nrm = 0;
    for (int r = k; r < rows; ++r) nrm += A[3 * r + k] * A[3 * r + k];
    nrm = sqrt(nrm);
    if (nrm == 0.0) return -1;
    const double alpha = (A[3 * k + k] > 0) ? -nrm : nrm;
    double *v = malloc((size_t)rows * sizeof(double));
    double vn = 0;
    for (int r = k; r < rows; ++r) { v[r] = A[3 * r + k]; if (r == k) v[r] -= alpha; vn += v[r] * v[r]; }
    if (vn > 0) {
      for (int cc = k; cc < 3; ++cc) {
        double d = 0;
        for (int r = k; r < rows; ++r) d += v[r] * A[3 * r + cc];
        d = 2.0 * d / vn;
        for (int r = k; r < rows; ++r) A[3 * r + cc] -= d * v[r];
      }
      double d = 0;
      for (int r = k; r < rows; ++r) d += v[r] * b[r];
      d = 2.0 * d / vn;
      for (int r = k; r < rows; ++r) b[r] -= d * v[r];
    }
    free(v);
  }
  for (int k = 2; k >= 0; --k) {
    double sacc = b[k];
    for (int cc = k + 1; cc < 3; ++cc) sacc -= A[3 * k + cc] * x[cc];
    if (A[3 * k + k] == 0.0) return -1;
    x[k] = sacc / A[3 * k + k];
  }
  return 0;
}

/* Registrator::refineAxis (mvr/src/registrator.cpp:402-455).  poses: n registered views' 4x4 poses, column-major,
 * column-vector convention [R t] (the transpose of the osg::Matrix the reference reads: its A(i*3+j, k) =
 * matrices[i](k, j) - delta_jk is R - I).  Axis: (R_i - I) x = 0 for all i, plus the row u + v + w = 1, least squares,
 * then normalised as an osg::Vec3 (float).  Pivot: (R_i - I) p = -t_i for all i, plus the row p_y = pivot_y (the
 * current pivot's y, a float), least squares, stored as floats.  Returns 0, -1 when n == 0 or a system is rank
 * deficient (outputs untouched). */
API int orc_refine_axis(int n, const double *poses, float pivot_y, float axis_out[3], float pivot_out[3])
{
  if (n <= 0) return -1;
  const int rows = 3 * n + 1;
  double *A = malloc((size_t)rows * 3 * sizeof(double)), *b = malloc((size_t)rows * sizeof(double));
  double x[3] = { 0, 0, 0 };
  int rc = 0;
  for (int pass = 0; pass < 2 && rc == 0; ++pass) {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < 3; ++j) {
        for (int k = 0; k < 3; ++k) A[3 * (3 * i + j) + k] = poses[16 * i + j + 4 * k] - ((j == k) ? 1.0 : 0.0);
        b[3 * i + j] = pass ? -poses[16 * i + 12 + j] : 0.0;
      }
    if (pass == 0) { A[3 * (rows - 1)] = 1; A[3 * (rows - 1) + 1] = 1; A[3 * (rows - 1) + 2] = 1; b[rows - 1] = 1; }
    else { A[3 * (rows - 1)] = 0; A[3 * (rows - 1) + 1] = 1; A[3 * (rows - 1) + 2] = 0; b[rows - 1] = (double)pivot_y; }
    rc = lstsq3(A, b, rows, x);
    if (rc) break;
    if (pass == 0) {
      /* osg::Vec3 normal(x0,x1,x2); normal.normalize(): float components, float norm */
      float v[3] = { (float)x[0], (float)x[1], (float)x[2] };
      float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      if (nrm > 0.0f) { float inv = 1.0f / nrm; v[0] *= inv; v[1] *= inv; v[2] *= inv; }
      axis_out[0] = v[0]; axis_out[1] = v[1]; axis_out[2] = v[2];
    } else {
      pivot_out[0] = (float)x[0]; pivot_out[1] = (float)x[1]; pivot_out[2] = (float)x[2];
    }
  }
  free(A); free(b);
  return rc;
}
