// include/mvr/registration.hpp -- drop-in C++ shim over the C-ABI (mvr_hip.h).
//
// Source-compatible stand-ins for the three PCL classes the reference's
// Registrator drives:
//   pcl::IterativeClosestPoint<PointXYZ,PointXYZ>          mvr/include/registrator.h:91,
//       call sites mvr/src/registrator.cpp:551-576, 768-777, 901-923, 1012-1015, 1024-1025
//   pcl::registration::CorrespondenceEstimation<P,P,float> registrator.cpp:496-502, 644-649
//   pcl::registration::LUM<PointXYZ>                       registrator.cpp:627-658
// Same method names, argument meaning and error behaviour (align never throws
// for bad data: it reports through hasConverged(), as PCL does; HIP/runtime
// failures DO throw mvr::Error -- there is no CPU fallback to hide them).
// All arithmetic runs in libmvr_hip.so on the GPU (+ the tiny host solves).
// Like the reference, drive one object from one thread at a time.
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <string>

#include "../mvr_hip.h"
#include "types.hpp"

namespace mvr {

// One GPU context per thread, shared by the shim objects of that thread.
class Device {
 public:
  static Device &instance()
  {
    static thread_local Device d;
    return d;
  }
  mvr_ctx *ctx()
  {
    if (!ctx_) check(mvr_ctx_create(&ctx_, device_id()), "mvr_ctx_create (no usable GPU: there is no CPU fallback)");
    return ctx_;
  }
  static int &device_id() { static int id = 0; return id; }
  int acquire()
  {
    for (int s = 0; s < MVR_MAX_SLOTS; ++s) if (!used_[s]) { used_[s] = true; return s; }
    throw Error(MVR_E_NOMEM, "mvr: out of device cloud slots");
  }
  void release(int s) { if (s >= 0 && s < MVR_MAX_SLOTS) { used_[s] = false; if (ctx_) mvr_cloud_clear(ctx_, s); } }
  void check(int rc, const char *what)
  {
    if (rc != MVR_OK) throw Error(rc, std::string(what) + ": " + mvr_strerror(rc) + " (" + (ctx_ ? mvr_last_error(ctx_) : "") + ")");
  }
  template <typename PointT>
  void upload(int slot, const PointCloud<PointT> &c)
  {
    static_assert(sizeof(PointT) == 16, "16-byte XYZ points expected");
    check(mvr_cloud_upload(ctx(), slot, c.empty() ? nullptr : c.points[0].data, c.size(), sizeof(PointT)), "mvr_cloud_upload");
  }
  template <typename PointT>
  void download(int slot, PointCloud<PointT> &c)
  {
    size_t n = 0;
    check(mvr_cloud_size(ctx(), slot, &n), "mvr_cloud_size");
    c.resize(n);
    if (n) check(mvr_cloud_download(ctx(), slot, c.points[0].data, n, sizeof(PointT), &n), "mvr_cloud_download");
  }
  ~Device() { if (ctx_) mvr_ctx_destroy(ctx_); }

 private:
  Device() {}
  mvr_ctx *ctx_ = nullptr;
  bool used_[MVR_MAX_SLOTS] = {false};
};

struct SlotGuard {
  int s;
  SlotGuard() : s(Device::instance().acquire()) {}
  ~SlotGuard() { Device::instance().release(s); }
  SlotGuard(const SlotGuard &) = delete;
  SlotGuard &operator=(const SlotGuard &) = delete;
};

// ---------------------------------------------------------------- ICP
template <typename PointSource, typename PointTarget, typename Scalar = float>
class IterativeClosestPoint {
 public:
  typedef PointCloud<PointSource> PointCloudSource;
  typedef PointCloud<PointTarget> PointCloudTarget;
  typedef typename PointCloudSource::Ptr PointCloudSourcePtr;
  typedef typename PointCloudTarget::Ptr PointCloudTargetPtr;
  typedef Mat4<Scalar> Matrix4;

  IterativeClosestPoint()
  {
    // PCL defaults
    p_ = mvr_icp_params();     // zero every field, incl. the point_to_plane extension switch
    p_.use_reciprocal = 0; p_.max_corr_dist = std::sqrt(DBL_MAX); p_.max_iterations = 10;
    p_.transformation_epsilon = 0.0; p_.euclidean_fitness_eps = -DBL_MAX; p_.fma_dist = 0;
  }

  void setUseReciprocalCorrespondences(bool b) { p_.use_reciprocal = b ? 1 : 0; }
  bool getUseReciprocalCorrespondences() const { return p_.use_reciprocal != 0; }
  void setMaxCorrespondenceDistance(double d) { p_.max_corr_dist = d; }
  double getMaxCorrespondenceDistance() const { return p_.max_corr_dist; }
  void setMaximumIterations(int n) { p_.max_iterations = n; }
  int getMaximumIterations() const { return p_.max_iterations; }
  void setTransformationEpsilon(double e) { p_.transformation_epsilon = e; }
  double getTransformationEpsilon() const { return p_.transformation_epsilon; }
  void setEuclideanFitnessEpsilon(double e) { p_.euclidean_fitness_eps = e; }
  double getEuclideanFitnessEpsilon() const { return p_.euclidean_fitness_eps; }
  void setInputSource(const PointCloudSourcePtr &c) { input_ = c; }
  void setInputTarget(const PointCloudTargetPtr &c) { target_ = c; }
  PointCloudSourcePtr const getInputSource() const { return input_; }
  PointCloudTargetPtr const getInputTarget() const { return target_; }

  // align(output): `output` may be the very cloud handed to setInputSource
  // (registrator.cpp:920 `icp_.align(*source_)`): the input is uploaded first.
  void align(PointCloudSource &output)
  {
    Device &d = Device::instance();
    converged_ = false; final_ = Matrix4::Identity(); stats_ = mvr_icp_stats();
    if (!input_ || !target_) { state_ = MVR_CONV_NO_CORRESPONDENCES; return; }
    // the caller may have mutated either cloud since the last call (:576)
    d.upload(src_.s, *input_);
    d.upload(tgt_.s, *target_);
    float T[16];
    const int rc = mvr_icp_align(d.ctx(), src_.s, tgt_.s, out_.s, &p_, T, &stats_);
    if (rc != MVR_OK && rc != MVR_E_NOCORR) d.check(rc, "mvr_icp_align");
    if (rc == MVR_E_NOCORR)
      std::fprintf(stderr, "[mvr::IterativeClosestPoint::computeTransformation] Not enough correspondences found. "
                           "Relax your threshold parameters.\n");
    for (int k = 0; k < 16; ++k) final_.m[k] = (Scalar)T[k];
    converged_ = stats_.converged != 0;
    state_ = stats_.state;
    d.download(out_.s, output);
    aligned_once_ = true;
  }

  Matrix4 getFinalTransformation() const { return final_; }
  bool hasConverged() const { return converged_; }
  int getConvergenceState() const { return state_; }
  const mvr_icp_stats &getStats() const { return stats_; }

  // Registration::getFitnessScore(max_range): uses the input cloud and the
  // target as they were at the last align (the registration's own tree),
  // i.e. the clouds still resident in this object's device slots.
  double getFitnessScore(double max_range = DBL_MAX)
  {
    Device &d = Device::instance();
    if (!aligned_once_) { if (!input_ || !target_) return DBL_MAX; d.upload(src_.s, *input_); d.upload(tgt_.s, *target_); }
    else if (input_) d.upload(src_.s, *input_);       // PCL transforms *input_ as it is NOW (App. C.2)
    float T[16];
    for (int k = 0; k < 16; ++k) T[k] = (float)final_.m[k];
    double score = DBL_MAX;
    d.check(mvr_fitness(d.ctx(), src_.s, tgt_.s, T, max_range, p_.fma_dist, &score), "mvr_fitness");
    return score;
  }

 private:
  mvr_icp_params p_;
  mvr_icp_stats stats_ = mvr_icp_stats();
  PointCloudSourcePtr input_;
  PointCloudTargetPtr target_;
  Matrix4 final_;
  bool converged_ = false, aligned_once_ = false;
  int state_ = MVR_CONV_NOT;
  SlotGuard src_, tgt_, out_;
};

namespace registration {

// ------------------------------------------------- CorrespondenceEstimation
template <typename PointSource, typename PointTarget, typename Scalar = float>
class CorrespondenceEstimation {
 public:
  typedef typename PointCloud<PointSource>::Ptr PointCloudSourcePtr;
  typedef typename PointCloud<PointTarget>::Ptr PointCloudTargetPtr;
  void setInputSource(const PointCloudSourcePtr &c) { input_ = c; }
  void setInputTarget(const PointCloudTargetPtr &c) { target_ = c; }

  void determineReciprocalCorrespondences(Correspondences &out, double max_distance = DBL_MAX) { run(out, max_distance, 1); }
  void determineCorrespondences(Correspondences &out, double max_distance = DBL_MAX) { run(out, max_distance, 0); }

 private:
  void run(Correspondences &out, double max_distance, int reciprocal)
  {
    out.clear();
    if (!input_ || !target_ || input_->empty()) return;
    Device &d = Device::instance();
    d.upload(src_.s, *input_);
    d.upload(tgt_.s, *target_);
    const size_t ns = input_->size();
    std::vector<int32_t> q(ns), m(ns);
    std::vector<float> dd(ns);
    size_t n = 0;
    // sqrt(DBL_MAX)^2 == DBL_MAX: an unbounded search, as PCL's default
    const double md = std::min(max_distance, std::sqrt(DBL_MAX));
    d.check(mvr_correspondences(d.ctx(), src_.s, tgt_.s, md, reciprocal, 0, q.data(), m.data(), dd.data(), ns, &n),
            "mvr_correspondences");
    out.resize(n);
    for (size_t k = 0; k < n; ++k) out[k] = Correspondence(q[k], m[k], dd[k]);
  }
  PointCloudSourcePtr input_;
  PointCloudTargetPtr target_;
  SlotGuard src_, tgt_;
};

// ---------------------------------------------------------------------- LUM
template <typename PointT>
class LUM {
 public:
  typedef PointCloud<PointT> Cloud;
  typedef typename Cloud::Ptr PointCloudPtr;
  typedef size_t Vertex;

  size_t getNumVertices() const { return clouds_.size(); }
  void setMaxIterations(int n) { max_iterations_ = n; }
  int getMaxIterations() const { return max_iterations_; }
  void setConvergenceThreshold(float t) { threshold_ = t; }
  float getConvergenceThreshold() const { return threshold_; }

  Vertex addPointCloud(const PointCloudPtr &cloud, const Vector6f &pose = Vector6f())
  {
    clouds_.push_back(cloud);
    for (int k = 0; k < 6; ++k) poses_.push_back(clouds_.size() == 1 ? 0.0 : (double)pose(k));   // vertex 0 is the reference
    return clouds_.size() - 1;
  }
  PointCloudPtr getPointCloud(Vertex v) const { return v < clouds_.size() ? clouds_[v] : PointCloudPtr(); }
  void setCorrespondences(Vertex s, Vertex t, const CorrespondencesPtr &corrs)
  {
    if (s >= clouds_.size() || t >= clouds_.size() || s == t) {
      std::fprintf(stderr, "[mvr::registration::LUM::setCorrespondences] invalid vertices\n");
      return;
    }
    for (Edge &e : edges_) if (e.s == (int)s && e.t == (int)t) { e.corrs = corrs; return; }
    edges_.push_back(Edge{(int)s, (int)t, corrs});
  }
  CorrespondencesPtr getCorrespondences(Vertex s, Vertex t) const
  {
    for (const Edge &e : edges_) if (e.s == (int)s && e.t == (int)t) return e.corrs;
    return CorrespondencesPtr();
  }
  Vector6f getPose(Vertex v) const { Vector6f p; for (int k = 0; k < 6; ++k) p(k) = (float)poses_[6 * v + k]; return p; }

  // LUM::compute.  One GPU pass per edge reduces its
  // correspondences to raw second moments; the max_iterations_ linearised
  // solves then run on the host from those moments.
  void compute()
  {
    const int n = (int)clouds_.size();
    if (n < 2) { std::fprintf(stderr, "[mvr::registration::LUM::compute] The slam graph needs at least 2 vertices.\n"); return; }
    Device &d = Device::instance();
    std::vector<SlotGuard> slots(n);
    double origin[3] = {0, 0, 0};
    for (int v = 0; v < n; ++v) d.upload(slots[v].s, *clouds_[v]);
    if (!clouds_[0]->empty()) { const PointT &p = clouds_[0]->points[0]; origin[0] = p.x; origin[1] = p.y; origin[2] = p.z; }
    std::vector<mvr_pair_moments2_t> m2(edges_.size());
    std::vector<int> es(edges_.size()), et(edges_.size());
    for (size_t e = 0; e < edges_.size(); ++e) {
      const Correspondences &c = *edges_[e].corrs;
      std::vector<int32_t> q(c.size()), m(c.size());
      for (size_t k = 0; k < c.size(); ++k) { q[k] = c[k].index_query; m[k] = c[k].index_match; }
      es[e] = edges_[e].s; et[e] = edges_[e].t;
      d.check(mvr_pair_moments2_from_corr(d.ctx(), slots[es[e]].s, slots[et[e]].s, q.data(), m.data(), c.size(), origin, &m2[e]),
              "mvr_pair_moments2_from_corr");
    }
    int iters = 0;
    const int rc = mvr_lum_compute(n, (int)edges_.size(), es.data(), et.data(), m2.data(), max_iterations_,
                                   (double)threshold_, poses_.data(), &iters);
    if (rc != MVR_OK) std::fprintf(stderr, "[mvr::registration::LUM::compute] %s\n", mvr_strerror(rc));
  }

  Affine3f getTransformation(Vertex v) const
  {
    double T[16];
    mvr_pose_to_mat4(&poses_[6 * v], T);
    Affine3f a;
    for (int k = 0; k < 16; ++k) a.mat.m[k] = (float)T[k];
    return a;
  }

 private:
  struct Edge { int s, t; CorrespondencesPtr corrs; };
  std::vector<PointCloudPtr> clouds_;
  std::vector<double> poses_;
  std::vector<Edge> edges_;
  int max_iterations_ = 5;      // PCL default; the reference sets 16 (registrator.cpp:623,653)
  float threshold_ = 0.0f;
};

}  // namespace registration
}  // namespace mvr

#ifdef MVR_ALIAS_PCL
namespace pcl = mvr;   // lets `pcl::IterativeClosestPoint<...>` etc. resolve to the shim
#endif
