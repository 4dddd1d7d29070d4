// include/mvr/registrator.hpp -- the reference's registration DRIVER, re-hosted
// on the shim classes of registration.hpp (so it runs on the GPU path).
//
// Mirrors, call for call, the non-GUI part of class Registrator
// (mvr/include/registrator.h:40-59, mvr/src/registrator.cpp):
//   getRotationMatrix(angle)                       :331-342
//   computeError(object)                           :466-515
//   registrationICP(max_it, max_d, obj[, repeat])  :517-588
//   registrationLUM(seg, max_it, max_d, obj)       :611-678
//   automaticRegistration / automaticRegistrationICP / automaticRefineTransformation /
//   refineTransformation                           :746-842, :877-990, :1008-1030
//   refineAxis(object)                             :402-455 (+ math_solvers::least_squares, math_solvers.cpp:12-38)
// and of class PointCloud (mvr/src/point_cloud.cpp): getTransformedPoints
// :290-303, initRotation :400-413, set/getMatrix, isShown/isRegistered.
// The Qt/OSG/file-tree plumbing (FileSystemModel, QtConcurrent, draggers,
// rendering, dialogs) is out of scope: `TurntableModel` is an in-memory stand-in
// for FileSystemModel::getPointCloud(object, view).
#pragma once

#include <cstdio>
#include <cstring>
#include <iostream>
#include <utility>
#include <vector>

#include "registration.hpp"

namespace mvr {

typedef PointXYZ PCLPoint;                       // mvr/include/types.h:14
typedef PointCloud<PCLPoint> PCLPointCloud;      // mvr/include/types.h:17

class Registrator;

// One scan: raw points + its pose (osg::Matrix of the reference's
// osg::MatrixTransform base) + the flags the driver consults.
class ScanCloud {
 public:
  PCLPointCloud points;
  int view = 0;
  bool shown = true;
  bool registered = false;

  size_t size() const { return points.size(); }
  const RowMatrixd &getMatrix() const { return matrix_; }
  void setMatrix(const RowMatrixd &m) { matrix_ = m; }
  bool isShown() const { return shown; }
  bool isRegistered() const { return registered; }
  void setRegisterState(bool r) { registered = r; }
  int getView() const { return view; }

  // point_cloud.cpp:290-303 -- (x,y,z) * M in double, rounded to float.  K1 on the GPU.
  void getTransformedPoints(PCLPointCloud &out) const
  {
    Device &d = Device::instance();
    SlotGuard s;
    d.upload(s.s, points);
    d.check(mvr_cloud_transform(d.ctx(), s.s, s.s, matrix_.asColumnMajorColumnVector()), "mvr_cloud_transform");
    d.download(s.s, out);
  }
  void initRotation(const Registrator &r);        // point_cloud.cpp:400-413

  // point_cloud.cpp:423-465: drop the connected components (points linked when <= triangle_length apart -- the
  // reference's short Delaunay edges give the same components) with fewer than segment_threshold points; the
  // survivors are stored component after component, as the reference's denoised_cloud.  Returns the noise count.
  size_t denoise(int segment_threshold, double triangle_length)
  {
    Device &d = Device::instance();
    SlotGuard s;
    const size_t before = points.size();
    d.upload(s.s, points);
    size_t kept = 0, comps = 0;
    d.check(mvr_cloud_denoise(d.ctx(), s.s, segment_threshold, triangle_length, &kept, &comps, nullptr), "mvr_cloud_denoise");
    d.download(s.s, points);
    return before - kept;
  }

  // point_cloud.cpp:305-326 / :328-347 -- `transformation.txt`: the column-vector
  // 4x4 printed row by row (matrix(j,i), i outer), "%lf " per element, one row per
  // line.  6 decimals: poses round-trip to 1e-6 only.
  bool loadTransformation(const std::string &filename)
  {
    FILE *file = std::fopen(filename.c_str(), "r");
    if (file == NULL) return false;
    RowMatrixd matrix;
    bool ok = true;
    for (int i = 0; i < 4 && ok; ++i)
      for (int j = 0; j < 4 && ok; ++j) {
        double element;
        ok = std::fscanf(file, "%lf", &element) == 1;
        if (ok) matrix(j, i) = element;
      }
    std::fclose(file);
    if (ok) setMatrix(matrix);
    return ok;
  }
  bool saveTransformation(const std::string &filename) const
  {
    FILE *file = std::fopen(filename.c_str(), "w");
    if (file == NULL) return false;
    for (int i = 0; i < 4; ++i) {
      for (int j = 0; j < 4; ++j) std::fprintf(file, "%lf ", matrix_(j, i));
      std::fprintf(file, "\n");
    }
    std::fclose(file);
    return true;
  }

 private:
  RowMatrixd matrix_;
};

// stand-in for FileSystemModel::getPointCloud(object, view): views 0..V-1
struct TurntableModel {
  std::vector<ScanCloud> views;
  ScanCloud &getPointCloud(int /*object*/, int view) { return views.at((size_t)view); }
  int numViews() const { return (int)views.size(); }
};

struct AlignLog { int view; Matrix4f T; int n_corr; double mse; int iterations; double fitness; bool has_fitness; };

class Registrator {
 public:
  explicit Registrator(TurntableModel *model) : model_(model) {}

  // pivot / axis are osg::Vec3 (float) in the reference (registrator.h)
  void setPivotPoint(double x, double y, double z) { pivot_[0] = (float)x; pivot_[1] = (float)y; pivot_[2] = (float)z; }
  void setAxisNormal(double x, double y, double z) { axis_[0] = (float)x; axis_[1] = (float)y; axis_[2] = (float)z; }
  const float *getPivotPoint() const { return pivot_; }
  const float *getAxisNormal() const { return axis_; }

  // registrator.cpp:258-274 / :294-308 -- `axis.txt`: pivot "x y z" then axis "nx ny nz", "%f"
  bool load(const std::string &filename)
  {
    FILE *file = std::fopen(filename.c_str(), "r");
    if (file == NULL) return false;
    double x, y, z, nx, ny, nz;
    const bool ok = std::fscanf(file, "%lf %lf %lf", &x, &y, &z) == 3 && std::fscanf(file, "%lf %lf %lf", &nx, &ny, &nz) == 3;
    std::fclose(file);
    if (ok) { setPivotPoint(x, y, z); setAxisNormal(nx, ny, nz); }
    return ok;
  }
  bool save(const std::string &filename) const
  {
    FILE *file = std::fopen(filename.c_str(), "w");
    if (file == NULL) return false;
    std::fprintf(file, "%f %f %f\n", pivot_[0], pivot_[1], pivot_[2]);
    std::fprintf(file, "%f %f %f\n", axis_[0], axis_[1], axis_[2]);
    std::fclose(file);
    return true;
  }

  // registrator.cpp:331-342
  RowMatrixd getRotationMatrix(double angle) const
  {
    RowMatrixd matrix = RowMatrixd::identity();
    matrix = matrix * RowMatrixd::translate(-pivot_[0], -pivot_[1], -pivot_[2]);
    matrix = matrix * RowMatrixd::rotate(angle, axis_[0], axis_[1], axis_[2]);
    matrix = matrix * RowMatrixd::translate(pivot_[0], pivot_[1], pivot_[2]);
    return matrix;
  }

  // point_cloud.cpp:409 generalised from 12 views / 30 degrees
  double viewAngle(int view) const { return mvr_turntable_angle(view, model_->numViews()); }

  // registrator.cpp:466-515: ring pairs (i,i+1) of shown views plus (0, V-1);
  // returns per pair the reciprocal correspondences (the reference leaves the
  // visualisation of them commented out, :504-511).
  std::vector<std::pair<std::pair<int, int>, CorrespondencesPtr> > computeError(int object, double distance_threshold)
  {
    const int V = model_->numViews();
    std::vector<bool> shown_flag(V, false);
    shown_flag[0] = true;
    for (int i = 1; i < V; ++i) {
      ScanCloud &pc = model_->getPointCloud(object, i);
      shown_flag[i] = pc.isShown();
      if (shown_flag[i]) pc.initRotation(*this);
    }
    std::vector<std::pair<int, int> > neighbor_pairs;
    for (int i = 0; i < V - 1; ++i) if (shown_flag[i] && shown_flag[i + 1]) neighbor_pairs.push_back(std::make_pair(i, i + 1));
    if (shown_flag[0] && shown_flag[V - 1]) neighbor_pairs.push_back(std::make_pair(0, V - 1));
    std::vector<std::pair<std::pair<int, int>, CorrespondencesPtr> > result;
    PCLPointCloud::Ptr source(new PCLPointCloud), target(new PCLPointCloud);
    for (size_t i = 0; i < neighbor_pairs.size(); ++i) {
      model_->getPointCloud(object, neighbor_pairs[i].first).getTransformedPoints(*source);
      model_->getPointCloud(object, neighbor_pairs[i].second).getTransformedPoints(*target);
      registration::CorrespondenceEstimation<PCLPoint, PCLPoint, float> correspondence_estimation;
      correspondence_estimation.setInputSource(source);
      correspondence_estimation.setInputTarget(target);
      CorrespondencesPtr correspondences(new Correspondences);
      correspondence_estimation.determineReciprocalCorrespondences(*correspondences, distance_threshold);
      result.push_back(std::make_pair(neighbor_pairs[i], correspondences));
    }
    return result;
  }

  // registrator.cpp:517-524
  void registrationICP(int max_iterations, double max_distance, int object, int repeat_times)
  {
    for (int i = 0; i < repeat_times; i++) registrationICP(max_iterations, max_distance, object);
  }

  // registrator.cpp:526-588
  void registrationICP(int max_iterations, double max_distance, int object)
  {
    const int V = model_->numViews();
    std::vector<ScanCloud *> point_clouds;
    for (int i = 1; i < V / 2; ++i) {
      ScanCloud &front_cloud = model_->getPointCloud(object, i);
      if (front_cloud.isShown()) point_clouds.push_back(&front_cloud);
      ScanCloud &back_cloud = model_->getPointCloud(object, V - i);
      if (back_cloud.isShown()) point_clouds.push_back(&back_cloud);
    }
    ScanCloud &center_cloud = model_->getPointCloud(object, V / 2);
    if (center_cloud.isShown()) point_clouds.push_back(&center_cloud);
    if (point_clouds.empty()) return;

    for (size_t i = 0; i < point_clouds.size(); ++i) point_clouds[i]->initRotation(*this);

    PCLPointCloud::Ptr source(new PCLPointCloud);
    PCLPointCloud::Ptr target(new PCLPointCloud);

    IterativeClosestPoint<PCLPoint, PCLPoint> icp;
    icp.setUseReciprocalCorrespondences(true);
    icp.setMaxCorrespondenceDistance(max_distance);
    icp.setMaximumIterations(max_iterations);
    icp.setTransformationEpsilon(0.000001);
    icp.setEuclideanFitnessEpsilon(64);

    model_->getPointCloud(object, 0).getTransformedPoints(*target);
    for (size_t i = 0, i_end = point_clouds.size(); i < i_end; ++i) {
      point_clouds[i]->getTransformedPoints(*source);
      icp.setInputSource(source);
      icp.setInputTarget(target);
      PCLPointCloud transformed_source;
      icp.align(transformed_source);

      AlignLog entry{point_clouds[i]->getView(), icp.getFinalTransformation(), icp.getStats().n_corr, icp.getStats().mse,
                     icp.getStats().iterations, 0.0, false};
      if (i == i_end - 1) {
        entry.fitness = icp.getFitnessScore(); entry.has_fitness = true;
        if (verbose) std::cout << "i:" << i << " " << entry.fitness << std::endl;
      }
      log.push_back(entry);
      RowMatrixd result_matrix = PclMatrixCaster<RowMatrixd>(icp.getFinalTransformation());
      point_clouds[i]->setMatrix(point_clouds[i]->getMatrix() * result_matrix);

      *target += transformed_source;
    }
  }

  // registrator.cpp:611-678 (without saveRegisteredPoints / refineAxis / expire)
  void registrationLUM(int /*segment_threshold*/, int max_iterations, double max_distance, int object)
  {
    const int V = model_->numViews();
    for (int view = 0; view < V; ++view) {
      ScanCloud &pc = model_->getPointCloud(object, view);
      pc.initRotation(*this);
      pc.setRegisterState(true);
    }
    int lum_max_iterations = 16;
    int outer_loop_num = std::max(1, max_iterations / lum_max_iterations);
    for (int loop = 0; loop < outer_loop_num; ++loop) {
      registration::LUM<PCLPoint> lum;
      for (int i = 0; i < V; ++i) {
        ScanCloud &pc = model_->getPointCloud(object, i);
        pc.initRotation(*this);
        PCLPointCloud::Ptr transformed_cloud(new PCLPointCloud);
        pc.getTransformedPoints(*transformed_cloud);
        lum.addPointCloud(transformed_cloud);
      }
      lum_ncorr.clear();
      for (int i = 0; i < V; ++i) {
        int source_idx = i;
        int target_idx = (i == V - 1) ? (0) : (i + 1);
        registration::CorrespondenceEstimation<PCLPoint, PCLPoint, float> correspondence_estimation;
        correspondence_estimation.setInputSource(lum.getPointCloud(source_idx));
        correspondence_estimation.setInputTarget(lum.getPointCloud(target_idx));
        CorrespondencesPtr correspondences(new Correspondences);
        correspondence_estimation.determineReciprocalCorrespondences(*correspondences, max_distance);
        lum.setCorrespondences(source_idx, target_idx, correspondences);
        lum_ncorr.push_back((int)correspondences->size());
      }
      lum.setMaxIterations(lum_max_iterations);
      lum.compute();
      for (int i = 0; i < V; ++i) {
        Affine3f transformation = lum.getTransformation(i);
        RowMatrixd osg_transformation = PclMatrixCaster<RowMatrixd>(Matrix4f(transformation.data()));
        ScanCloud &pc = model_->getPointCloud(object, i);
        pc.setMatrix(pc.getMatrix() * osg_transformation);
        pc.setRegisterState(true);
      }
    }
  }

  // The same outer passes (registrator.cpp:611-678) with everything DEVICE-RESIDENT -- the form the
  // reference's loop takes when no correspondence list has to reach the host: the scans are uploaded
  // once; per outer pass all views are posed and re-indexed by one launch each
  // (mvr_cloud_transform_batch), the V ring pairs are searched concurrently on worker HIP streams and
  // reduced to raw second moments (mvr_pair_moments2_batch), and the per-pair solve + LUM + pose
  // update run on the host from those V x 31 doubles (mvr_ring_host_step).  Same poses as
  // registrationLUM up to float rounding of the intermediate PCL-style transforms.
  void registrationLUMDevice(int max_iterations, double max_distance, int object)
  {
    const int V = model_->numViews();
    if (V < 2) return;
    Device &d = Device::instance();
    std::vector<SlotGuard> raw((size_t)V), posed((size_t)V);
    std::vector<int> raw_s((size_t)V), posed_s((size_t)V), es((size_t)V), et((size_t)V);
    double origin[3] = {0.0, 0.0, 0.0};
    for (int v = 0; v < V; ++v) {
      ScanCloud &pc = model_->getPointCloud(object, v);
      d.upload(raw[v].s, pc.points);
      raw_s[v] = raw[v].s; posed_s[v] = posed[v].s;
      es[v] = v; et[v] = (v == V - 1) ? 0 : v + 1;
    }
    if (!model_->getPointCloud(object, 0).points.empty()) {
      const PCLPoint &p = model_->getPointCloud(object, 0).points.points[0];
      origin[0] = p.x; origin[1] = p.y; origin[2] = p.z;
    }
    const int lum_max_iterations = 16;
    const int outer_loop_num = std::max(1, max_iterations / lum_max_iterations);
    std::vector<double> poses((size_t)V * 16), rows((size_t)V * 32), pair_n((size_t)V), pair_mse((size_t)V), lum_pose((size_t)V * 6);
    for (int loop = 0; loop < outer_loop_num; ++loop) {
      for (int v = 0; v < V; ++v) {
        ScanCloud &pc = model_->getPointCloud(object, v);
        pc.initRotation(*this);
        pc.setRegisterState(true);
        std::memcpy(&poses[(size_t)v * 16], pc.getMatrix().asColumnMajorColumnVector(), 16 * sizeof(double));
      }
      // posing, reciprocal correspondences + moments of every ring edge, table copy and the LUM solve: one native call
      int iters = 0;
      d.check(mvr_ring_step(d.ctx(), V, posed_s.data(), raw_s.data(), V, es.data(), et.data(), max_distance, 1, 0, origin,
                            lum_max_iterations, poses.data(), lum_pose.data(), nullptr, pair_n.data(), pair_mse.data(), &iters,
                            rows.data(), nullptr), "mvr_ring_step");
      lum_ncorr.clear();
      for (int e = 0; e < V; ++e) lum_ncorr.push_back((int)pair_n[e]);
      for (int v = 0; v < V; ++v) {
        RowMatrixd m;
        std::memcpy(&m(0, 0), &poses[(size_t)v * 16], 16 * sizeof(double));
        ScanCloud &pc = model_->getPointCloud(object, v);
        pc.setMatrix(m);
        pc.setRegisterState(true);
      }
    }
  }

  // registrator.cpp:1020-1030 / :1008-1018: `icp_.align(*source_)` with the
  // output aliasing the input, pose accumulated per repeat.
  void refineTransformation(int repeat_times, int source_index)
  {
    for (int i = 0; i < repeat_times; i++) {
      icp_.align(*source_);
      RowMatrixd result_matrix = PclMatrixCaster<RowMatrixd>(icp_.getFinalTransformation());
      point_clouds_[source_index]->setMatrix(point_clouds_[source_index]->getMatrix() * result_matrix);
      log.push_back(AlignLog{point_clouds_[source_index]->getView(), icp_.getFinalTransformation(), icp_.getStats().n_corr,
                             icp_.getStats().mse, icp_.getStats().iterations, 0.0, false});
    }
  }

  // The evident intent of automaticRegistration (:746-842) + automaticRegistrationICP
  // (:877-990): add the views one at a time, register each new view against the
  // merged target of all earlier ones with `repeat_times` in-place aligns, append
  // it.  (The original indexes point_clouds_ out of bounds for view >= 2 and
  // re-registers earlier views cumulatively -- SURVEY App. C.1; never calls
  // setTransformationEpsilon -- App. C.3.)
  void automaticRegistration(int object, int max_iterations, int repeat_times, double max_distance,
                             double euclidean_fitness_epsilon)
  {
    const int V = model_->numViews();
    if (!target_) target_.reset(new PCLPointCloud);
    if (!source_) source_.reset(new PCLPointCloud);
    model_->getPointCloud(object, 0).getTransformedPoints(*target_);
    point_clouds_.clear();
    for (int view_number = 1; view_number < V; ++view_number) {
      ScanCloud &pc = model_->getPointCloud(object, view_number);
      point_clouds_.push_back(&pc);
      const int source_index = (int)point_clouds_.size() - 1;
      pc.initRotation(*this);
      pc.setRegisterState(true);
      icp_.setUseReciprocalCorrespondences(true);
      icp_.setMaxCorrespondenceDistance(max_distance);
      icp_.setMaximumIterations(max_iterations);
      icp_.setEuclideanFitnessEpsilon(euclidean_fitness_epsilon);
      pc.getTransformedPoints(*source_);
      icp_.setInputSource(source_);
      icp_.setInputTarget(target_);
      refineTransformation(repeat_times, source_index);
      *target_ += *source_;
    }
  }

  // registrator.cpp:402-455: least-squares turntable axis from the registered
  // poses: (R^T - I) x = 0 with u+v+w = 1, then pivot from (R^T - I) p = -t with
  // p_y pinned.  math_solvers::least_squares (LAPACK dgels) -> normal equations here.
  void refineAxis(int object)
  {
    const int V = model_->numViews();
    std::vector<RowMatrixd> matrices;
    for (int i = 1; i < V; ++i) {
      ScanCloud &pc = model_->getPointCloud(object, i);
      if (!pc.isRegistered()) continue;
      matrices.push_back(pc.getMatrix());
    }
    if (matrices.empty()) return;
    const size_t rows = 3 * matrices.size() + 1;
    std::vector<double> A(rows * 3, 0.0), b(rows, 0.0);
    for (size_t i = 0; i < matrices.size(); ++i)
      for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) A[(i * 3 + j) * 3 + k] = matrices[i](k, j) - ((j == k) ? 1.0 : 0.0);
    const size_t idx = 3 * matrices.size();
    A[idx * 3 + 0] = 1; A[idx * 3 + 1] = 1; A[idx * 3 + 2] = 1; b[idx] = 1;
    double x[3];
    if (!leastSquares3(A, b, x)) return;
    const double n = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    setAxisNormal(x[0] / n, x[1] / n, x[2] / n);
    for (size_t i = 0; i < matrices.size(); ++i) for (int j = 0; j < 3; ++j) b[i * 3 + j] = -matrices[i](3, j);
    A[idx * 3 + 0] = 0; A[idx * 3 + 1] = 1; A[idx * 3 + 2] = 0; b[idx] = pivot_[1];
    if (!leastSquares3(A, b, x)) return;
    setPivotPoint(x[0], x[1], x[2]);
  }

  std::vector<AlignLog> log;       // one entry per align (what the reference prints / writes to fitness_scores.txt)
  std::vector<int> lum_ncorr;      // correspondences per ring edge of the last LUM pass
  bool verbose = false;

 private:
  static bool leastSquares3(const std::vector<double> &A, const std::vector<double> &b, double x[3])
  {
    double N[9] = {0}, r[3] = {0};
    const size_t rows = b.size();
    for (size_t i = 0; i < rows; ++i)
      for (int j = 0; j < 3; ++j) {
        r[j] += A[i * 3 + j] * b[i];
        for (int k = 0; k < 3; ++k) N[3 * j + k] += A[i * 3 + j] * A[i * 3 + k];
      }
    const double det = N[0] * (N[4] * N[8] - N[5] * N[7]) - N[1] * (N[3] * N[8] - N[5] * N[6]) + N[2] * (N[3] * N[7] - N[4] * N[6]);
    if (det == 0.0) return false;
    const double inv[9] = {(N[4] * N[8] - N[5] * N[7]) / det, (N[2] * N[7] - N[1] * N[8]) / det, (N[1] * N[5] - N[2] * N[4]) / det,
                           (N[5] * N[6] - N[3] * N[8]) / det, (N[0] * N[8] - N[2] * N[6]) / det, (N[2] * N[3] - N[0] * N[5]) / det,
                           (N[3] * N[7] - N[4] * N[6]) / det, (N[1] * N[6] - N[0] * N[7]) / det, (N[0] * N[4] - N[1] * N[3]) / det};
    for (int j = 0; j < 3; ++j) x[j] = inv[3 * j] * r[0] + inv[3 * j + 1] * r[1] + inv[3 * j + 2] * r[2];
    return true;
  }

  TurntableModel *model_;
  float pivot_[3] = {0, 0, 0};
  float axis_[3] = {0, 0, 1};
  // members of the reference's Registrator (registrator.h:88-93)
  std::vector<ScanCloud *> point_clouds_;
  PCLPointCloud::Ptr source_, target_;
  IterativeClosestPoint<PCLPoint, PCLPoint> icp_;
};

// point_cloud.cpp:400-413
inline void ScanCloud::initRotation(const Registrator &r)
{
  if (!getMatrix().isIdentity()) return;
  if (view == 0) return;
  setMatrix(r.getRotationMatrix(r.viewAngle(view)));
}

}  // namespace mvr
