// tests/cxx/call_surface.hpp -- TEST INFRASTRUCTURE: the PCL-named call surface of include/mvr/registration.hpp, exercised.
//
// The drop-in claim of the shim is about SIGNATURES AND BEHAVIOUR AT THE CALL SITES: every member function the reference's
// Registrator calls on pcl::IterativeClosestPoint, pcl::registration::CorrespondenceEstimation and pcl::registration::LUM
// (mvr/src/registrator.cpp:496-502, 551-576, 627-658, 768-777, 901-923, 1012-1015, 1024-1025) must exist under
// the same name, take the same arguments and react the same way -- an output cloud that aliases the input, a target
// that grows by operator+= between aligns, LUM vertices handed back by getPointCloud.  This header drives those members
// in the reference's ORDER OF CALLS through four small flows written for this test suite (they are not the reference's
// function bodies: the loops of the reference are restated, independently, in tests/ref_driver.py, which is what the
// results are compared with).  The product's own drivers are the device-resident ones of include/mvr/registrator.hpp.
#pragma once

#include <functional>
#include <iostream>

#include "mvr/registrator.hpp"

namespace surface {

using namespace mvr;
typedef IterativeClosestPoint<PCLPoint, PCLPoint> Icp;
typedef registration::CorrespondenceEstimation<PCLPoint, PCLPoint, float> Matcher;

// the five knobs a caller of align() can turn, applied through the five setters the reference uses
struct IcpKnobs {
  double max_distance = 4.0;
  int max_iterations = 10;
  bool with_transformation_epsilon = true;      // (the reference's incremental mode never sets it: it stays at PCL's default 0)
  double transformation_epsilon = 1e-6;
  double fitness_epsilon = 64.0;
  void configure(Icp &icp) const
  {
    icp.setUseReciprocalCorrespondences(true);
    icp.setMaxCorrespondenceDistance(max_distance);
    icp.setMaximumIterations(max_iterations);
    if (with_transformation_epsilon) icp.setTransformationEpsilon(transformation_epsilon);
    icp.setEuclideanFitnessEpsilon(fitness_epsilon);
  }
};

struct RingResidual { int a, b; CorrespondencesPtr pairs; };

class CallSurface : public mvr::Registrator {
 public:
  explicit CallSurface(TurntableModel *scans) : mvr::Registrator(scans), scans_(scans) {}

  // ---- flow 1: reciprocal correspondences between neighbours on the ring (closing pair last), shown views only.
  //      members: CorrespondenceEstimation::setInputSource / setInputTarget / determineReciprocalCorrespondences
  std::vector<RingResidual> ringResiduals(double threshold)
  {
    const int count = scans_->numViews();
    auto visible = [&](int v) { return v == 0 || view(v).isShown(); };
    for (int v = 1; v < count; ++v) if (visible(v)) view(v).initRotation(*this);
    std::vector<RingResidual> out;
    auto match = [&](int a, int b) {
      if (!visible(a) || !visible(b)) return;
      PCLPointCloud::Ptr from = posedCopy(a), onto = posedCopy(b);
      Matcher matcher;
      matcher.setInputSource(from);
      matcher.setInputTarget(onto);
      RingResidual r{a, b, CorrespondencesPtr(new Correspondences)};
      matcher.determineReciprocalCorrespondences(*r.pairs, threshold);
      out.push_back(r);
    };
    for (int v = 0; v + 1 < count; ++v) match(v, v + 1);
    match(0, count - 1);
    return out;
  }

  // ---- flow 2: a sweep over the views from both ends of the turntable inwards, each aligned against a model that grows
  //      by the aligned view.  members: the five setters, setInputSource / setInputTarget, align(fresh cloud),
  //      getFinalTransformation, getFitnessScore (last view of a sweep), PointCloud::operator+=
  void growingTargetSweeps(const IcpKnobs &knobs, int sweeps)
  {
    const std::vector<int> visit = inwardOrder();
    if (visit.empty()) return;
    for (int s = 0; s < sweeps; ++s) {
      for (int v : visit) view(v).initRotation(*this);
      Icp icp;
      knobs.configure(icp);
      PCLPointCloud::Ptr model = posedCopy(0);
      for (size_t k = 0; k < visit.size(); ++k) {
        PCLPointCloud::Ptr moving = posedCopy(visit[k]);
        icp.setInputSource(moving);
        icp.setInputTarget(model);
        PCLPointCloud aligned;
        icp.align(aligned);
        note(visit[k], icp, k + 1 == visit.size());
        compose(visit[k], icp.getFinalTransformation());
        *model += aligned;
      }
    }
  }

  // ---- flow 3: global relaxation.  Per outer pass a fresh LUM graph gets every posed view as a vertex, every ring edge
  //      the reciprocal correspondences between the clouds THE GRAPH hands back, sixteen iterations, and every view the
  //      transformation of its vertex.  members: LUM::addPointCloud / getPointCloud / setCorrespondences /
  //      setMaxIterations / compute / getTransformation
  void ringRelaxation(int outer_passes, double max_distance)
  {
    const int count = scans_->numViews();
    for (int v = 0; v < count; ++v) { view(v).initRotation(*this); view(v).setRegisterState(true); }
    for (int pass = 0; pass < std::max(1, outer_passes); ++pass) {
      registration::LUM<PCLPoint> graph;
      for (int v = 0; v < count; ++v) { view(v).initRotation(*this); graph.addPointCloud(posedCopy(v)); }
      lum_ncorr.assign((size_t)count, 0);
      for (int v = 0; v < count; ++v) {
        const int next = (v + 1) % count;
        Matcher matcher;
        matcher.setInputSource(graph.getPointCloud(v));
        matcher.setInputTarget(graph.getPointCloud(next));
        CorrespondencesPtr found(new Correspondences);
        matcher.determineReciprocalCorrespondences(*found, max_distance);
        graph.setCorrespondences(v, next, found);
        lum_ncorr[(size_t)v] = (int)found->size();
      }
      graph.setMaxIterations(16);
      graph.compute();
      for (int v = 0; v < count; ++v) {
        const Affine3f moved = graph.getTransformation(v);
        compose(v, Matrix4f(moved.data()));
        view(v).setRegisterState(true);
      }
    }
  }

  // ---- flow 4: views added one at a time to a persistent model, each advanced IN PLACE by repeated aligns whose output
  //      cloud is the input cloud.  members: align(*source) with aliasing, getFinalTransformation per repeat,
  //      operator+= of the advanced source into the member target
  void addViewsOneByOne(const IcpKnobs &knobs, int repeats)
  {
    const int count = scans_->numViews();
    model_ = posedCopy(0);
    for (int v = 1; v < count; ++v) {
      view(v).initRotation(*this);
      view(v).setRegisterState(true);
      knobs.configure(member_icp_);
      moving_ = posedCopy(v);
      member_icp_.setInputSource(moving_);
      member_icp_.setInputTarget(model_);
      for (int r = 0; r < repeats; ++r) {
        member_icp_.align(*moving_);                       // the output IS the input
        compose(v, member_icp_.getFinalTransformation());
        note(v, member_icp_, false);
      }
      *model_ += *moving_;
    }
  }

  bool verbose = false;

 private:
  ScanCloud &view(int v) { return scans_->getPointCloud(0, v); }
  PCLPointCloud::Ptr posedCopy(int v)
  {
    PCLPointCloud::Ptr cloud(new PCLPointCloud);
    view(v).getTransformedPoints(*cloud);
    return cloud;
  }
  // 1, V-1, 2, V-2, ... meeting in the middle; hidden views are skipped
  std::vector<int> inwardOrder()
  {
    std::vector<int> order;
    int lo = 1, hi = scans_->numViews() - 1;
    auto take = [&](int v) { if (view(v).isShown()) order.push_back(v); };
    while (lo < hi) { take(lo++); take(hi--); }
    if (lo == hi) take(lo);
    return order;
  }
  // pose <- (pose, then T): the row-vector product the reference's osg matrices use
  void compose(int v, const Matrix4f &T)
  {
    const RowMatrixd step = PclMatrixCaster<RowMatrixd>(T);
    view(v).setMatrix(view(v).getMatrix() * step);
  }
  void note(int v, Icp &icp, bool with_fitness)
  {
    AlignLog e{view(v).getView(), icp.getFinalTransformation(), icp.getStats().n_corr, icp.getStats().mse, icp.getStats().iterations, 0.0, false};
    if (with_fitness) {
      e.fitness = icp.getFitnessScore();
      e.has_fitness = true;
      if (verbose) std::cout << "fitness of view " << v << ": " << e.fitness << std::endl;
    }
    log.push_back(e);
  }

  TurntableModel *scans_;
  PCLPointCloud::Ptr moving_, model_;      // (the reference keeps its incremental mode's source / target / icp as members too)
  Icp member_icp_;
};

}  // namespace surface
