// tests/cxx/reference_replay.hpp -- TEST INFRASTRUCTURE: the reference's own driver loops replayed on the shim.
//
// The drop-in claim of include/mvr/registration.hpp is that the reference's call sites compile and behave unchanged
// against the PCL-named shim classes.  This header is that demonstration: the non-GUI bodies of
//   Registrator::computeError            mvr/src/registrator.cpp:466-515
//   Registrator::registrationICP         :517-588
//   Registrator::registrationLUM         :611-664
//   automaticRegistration / refineTransformation (their evident intent)   :746-842, :877-990, :1008-1030
//   Registrator::refineAxis              :402-455
// with 12 -> V, osg::Matrix -> mvr::RowMatrixd and FileSystemModel -> mvr::TurntableModel, statement for statement --
// which is exactly why it lives under tests/ and not in the product include path: it is the reference's
// control flow, kept only to be run against the GPU path and compared with the oracle-driven restatement of the same
// loops (tests/test_gpu_shim.py).  The product's drivers are the device-resident ones of include/mvr/registrator.hpp.
#pragma once

#include <iostream>

#include "mvr/registrator.hpp"

namespace mvr_replay {

using namespace mvr;

class ReplayRegistrator : public mvr::Registrator {
 public:
  explicit ReplayRegistrator(TurntableModel *model) : mvr::Registrator(model), model_(model) {}

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
    A[idx * 3 + 0] = 0; A[idx * 3 + 1] = 1; A[idx * 3 + 2] = 0; b[idx] = getPivotPoint()[1];
    if (!leastSquares3(A, b, x)) return;
    setPivotPoint(x[0], x[1], x[2]);
  }

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
  // members of the reference's Registrator (registrator.h:88-93)
  std::vector<ScanCloud *> point_clouds_;
  PCLPointCloud::Ptr source_, target_;
  IterativeClosestPoint<PCLPoint, PCLPoint> icp_;
};

}  // namespace mvr_replay
