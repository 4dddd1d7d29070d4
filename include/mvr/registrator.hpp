// include/mvr/registrator.hpp -- the registration driver on top of the C-ABI: scan objects (raw points + pose +
// the reference's per-view text/PCD files) and a DEVICE-RESIDENT Registrator.
//
// What it stands for in the reference (mvr/src/registrator.cpp, mvr/src/point_cloud.cpp):
//   PointCloud::getTransformedPoints :290-303, initRotation :400-413, open :78-95, save :97-123,
//     load/saveTransformation :305-347, denoise :423-465                     -> ScanCloud
//   Registrator::getRotationMatrix :331-342, load/save (axis.txt) :258-308   -> Registrator (same formats)
//   registrationICP :517-588        -> Registrator::registrationICPDevice   (target grows in a device slot)
//   registrationLUM :611-664        -> Registrator::registrationLUMDevice   (all passes in one native call)
//   automaticRegistration :746-842, :877-990, :1008-1030 -> Registrator::automaticRegistrationDevice (in-place aligns, model grown in a slot)
//   computeError :466-515           -> Registrator::computeErrorDevice      (per-pair count + residual)
//   refineAxis :402-455             -> Registrator::refineAxis              (mvr_refine_axis)
//   saveRegisteredPoints :344-400, registration :719-744                    -> same names
// The Qt/OSG/file-tree plumbing (FileSystemModel, QtConcurrent, draggers, rendering, dialogs) is out of scope:
// `TurntableModel` is an in-memory stand-in for FileSystemModel::getPointCloud(object, view).
// The PCL-named call surface itself is exercised, in the reference's order of calls, by tests/cxx/call_surface.hpp.
#pragma once

#include <cstdio>
#include <cstring>
#include <iostream>
#include <utility>
#include <vector>

#include <memory>

#include "io.hpp"
#include "registration.hpp"

namespace mvr {

typedef PointXYZ PCLPoint;                       // mvr/include/types.h:14
typedef PointCloud<PCLPoint> PCLPointCloud;      // mvr/include/types.h:17

class Registrator;

// One scan: raw points + its pose (osg::Matrix of the reference's
// osg::MatrixTransform base) + the flags the driver consults.
class ScanCloud {
 public:
  PCLPointCloud points;          // what the registration path sees (point_cloud.cpp:297-299 strips the rest)
  io::RichCloud rich;            // the full PointXYZRGBNormal records when the scan came from a PCD file; may be empty
  int view = 0;
  bool shown = true;
  bool registered = false;

  size_t size() const { return points.size(); }
  // a counter the device-resident Registrator watches: bump it (touch) after editing `points` by hand
  unsigned long long revision() const { return revision_; }
  void touch() { ++revision_; }

  // PointCloud::open (point_cloud.cpp:78-95): the PCD file (ascii / binary / binary_compressed), the pose from the
  // transformation.txt beside it, registered = view 0 or a non-identity pose.  false: nothing changed.
  bool open(const std::string &filename)
  {
    io::RichCloud loaded;
    if (!io::loadPCDFile(filename, loaded)) return false;
    rich.swap(loaded);
    io::toXYZ(rich, points);
    filename_ = filename;
    matrix_.makeIdentity();
    loadTransformation(folder() + "/transformation.txt");
    registered = (view == 0) || !matrix_.isIdentity();
    touch();
    return true;
  }
  // PointCloud::save (point_cloud.cpp:97-123): "*.ply" = the XYZ points in the turntable's canonical frame (pivot at
  // the origin, axis along +z; the reference hard-codes its calibration there) as an ASCII PLY; anything else = the
  // rich cloud as binary_compressed PCD.
  bool save(const std::string &filename) const
  {
    if (filename.size() >= 3 && filename.compare(filename.size() - 3, 3, "ply") == 0) {
      const RowMatrixd transformation = RowMatrixd::translate(13.382786, -50.223461, -917.477600) *
                                        RowMatrixd::rotateFromTo(-0.054323, -0.814921, -0.577020, 0.0, 0.0, 1.0);
      PCLPointCloud canon;
      for (size_t i = 0; i < points.size(); ++i) {
        const PCLPoint &p = points.points[i];
        float o[3];
        for (int k = 0; k < 3; ++k)
          o[k] = (float)(transformation(0, k) * p.x + transformation(1, k) * p.y + transformation(2, k) * p.z + transformation(3, k));
        canon.push_back(PCLPoint(o[0], o[1], o[2]));
      }
      return io::savePLYFile(filename, canon);
    }
    if (rich.size() == points.size()) return io::savePCDFile(filename, rich, io::PCD_BINARY_COMPRESSED);
    io::RichCloud plain(points.size());
    for (size_t i = 0; i < points.size(); ++i) { plain[i].x = points.points[i].x; plain[i].y = points.points[i].y; plain[i].z = points.points[i].z; }
    return io::savePCDFile(filename, plain, io::PCD_BINARY_COMPRESSED);
  }
  const std::string &filename() const { return filename_; }
  std::string folder() const { const size_t k = filename_.find_last_of('/'); return k == std::string::npos ? std::string(".") : filename_.substr(0, k); }
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
    std::vector<uint32_t> index(before ? before : 1);
    d.check(mvr_cloud_denoise(d.ctx(), s.s, segment_threshold, triangle_length, &kept, &comps, index.data()), "mvr_cloud_denoise");
    d.download(s.s, points);
    if (rich.size() == before) {                   // colours and normals follow their points
      io::RichCloud r(kept);
      for (size_t k = 0; k < kept; ++k) r[k] = rich[index[k]];
      rich.swap(r);
    }
    touch();
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
  std::string filename_;
  unsigned long long revision_ = 0;
};

// stand-in for FileSystemModel::getPointCloud(object, view): views 0..V-1
struct TurntableModel {
  std::vector<ScanCloud> views;
  ScanCloud &getPointCloud(int /*object*/, int view) { return views.at((size_t)view); }
  int numViews() const { return (int)views.size(); }
};

struct AlignLog { int view; Matrix4f T; int n_corr; double mse; int iterations; double fitness; bool has_fitness; };

// The registration driver of this library: DEVICE-RESIDENT.  The scans of one object are uploaded once and stay in
// device slots for the whole session; a registration pass moves poses (V x 16 doubles) and per-pair sums
// (V x 32 doubles) across the bus, never points.  It covers the non-GUI duties of the reference's class Registrator
// (mvr/include/registrator.h:40-59) -- which pairs are registered in which order with which parameters, and how the
// result is composed into the views' poses -- without its PCL-shaped data flow (a host cloud rebuilt, re-uploaded and
// re-indexed at every align).  (tests/cxx/call_surface.hpp drives the PCL-named shim members in the reference's order of calls.)
class Registrator {
 public:
  explicit Registrator(TurntableModel *model) : model_(model) {}
  virtual ~Registrator() {}

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

  // ---- sequential registration against the growing model (what registrationICP computes, registrator.cpp:517-588):
  // views in the order 1, V-1, 2, V-2, ..., V/2 (shown ones only), each aligned to the merged cloud of view 0 and all
  // views before it, `repeat_times` sweeps.  Device form: the target grows IN PLACE in a reserved slot
  // (mvr_cloud_append), a view is posed from its resident raw scan (mvr_cloud_transform), aligned (mvr_icp_align: one
  // 4x4 comes back) and appended; the fitness score of the last view of a sweep is taken as the reference does.
  void registrationICPDevice(int max_iterations, double max_distance, int object, int repeat_times = 1)
  {
    const int V = model_->numViews();
    if (V < 2) return;
    Device &d = Device::instance();
    ensureResident(object);
    std::vector<int> order;
    for (int i = 1; i < V / 2; ++i) for (int v : {i, V - i}) if (model_->getPointCloud(object, v).isShown()) order.push_back(v);
    if (model_->getPointCloud(object, V / 2).isShown() && (order.empty() || order.back() != V / 2) && V / 2 >= 1) order.push_back(V / 2);
    if (order.empty()) return;
    size_t total = model_->getPointCloud(object, 0).size();
    for (int v : order) { model_->getPointCloud(object, v).initRotation(*this); total += model_->getPointCloud(object, v).size(); }
    mvr_icp_params p = mvr_icp_params();
    p.use_reciprocal = 1; p.max_corr_dist = max_distance; p.max_iterations = max_iterations;
    p.transformation_epsilon = 0.000001; p.euclidean_fitness_eps = 64;       // registrator.cpp:558-560
    SlotGuard target, source, moved;
    for (int sweep = 0; sweep < repeat_times; ++sweep) {
      d.check(mvr_cloud_transform(d.ctx(), target.s, raw_[0], model_->getPointCloud(object, 0).getMatrix().asColumnMajorColumnVector()), "mvr_cloud_transform");
      d.check(mvr_cloud_reserve(d.ctx(), target.s, total), "mvr_cloud_reserve");
      for (size_t k = 0; k < order.size(); ++k) {
        ScanCloud &pc = model_->getPointCloud(object, order[k]);
        d.check(mvr_cloud_transform(d.ctx(), source.s, raw_[(size_t)order[k]], pc.getMatrix().asColumnMajorColumnVector()), "mvr_cloud_transform");
        float T[16];
        mvr_icp_stats st = mvr_icp_stats();
        const int rc = mvr_icp_align(d.ctx(), source.s, target.s, moved.s, &p, T, &st);
        if (rc != MVR_OK && rc != MVR_E_NOCORR) d.check(rc, "mvr_icp_align");
        AlignLog entry{pc.getView(), Matrix4f(T), st.n_corr, st.mse, st.iterations, 0.0, false};
        if (k + 1 == order.size()) {
          d.check(mvr_fitness(d.ctx(), source.s, target.s, T, DBL_MAX, 0, &entry.fitness), "mvr_fitness");
          entry.has_fitness = true;
        }
        log.push_back(entry);
        pc.setMatrix(pc.getMatrix() * RowMatrixd(PclMatrixCaster<RowMatrixd>(entry.T)));      // pose <- T_icp o pose
        d.check(mvr_cloud_append(d.ctx(), target.s, moved.s), "mvr_cloud_append");
      }
    }
  }

  // ---- incremental registration (what automaticRegistration computes, registrator.cpp:746-842 with
  // automaticRegistrationICP :877-990 and refineTransformation :1008-1030; intent per SURVEY App. C.1 -- the original
  // indexes its view list out of bounds from the second view on and re-registers the earlier views cumulatively):
  // the views are added one at a time; a new view gets its turntable prior, is posed from its resident raw scan, then
  // advanced IN PLACE by `repeat_times` aligns against everything merged so far (the reference's `icp_.align(*source_)`
  // whose output aliases its input, :920 / :1012 / :1024: every repeat continues from the last, pose <- T_j o pose), and
  // is appended to the model where it stands.  The transformation epsilon is never set in this mode (it stays PCL's 0,
  // App. C.3); the fitness epsilon is the caller's (the dialog's default is 50).  Per repeat one 4x4 crosses the bus.
  // The logged fitness is the TRUE residual of the advanced source against the model (mvr_fitness with the identity),
  // not the reference's double-transformed number (App. C.2).
  void automaticRegistrationDevice(int object, int max_iterations, int repeat_times, double max_distance, double euclidean_fitness_epsilon,
                                   bool log_fitness = false)
  {
    const int V = model_->numViews();
    if (V < 2) return;
    Device &d = Device::instance();
    ensureResident(object);
    size_t total = 0;
    for (int v = 0; v < V; ++v) total += model_->getPointCloud(object, v).size();
    mvr_icp_params p = mvr_icp_params();
    p.use_reciprocal = 1; p.max_corr_dist = max_distance; p.max_iterations = max_iterations;
    p.transformation_epsilon = 0.0; p.euclidean_fitness_eps = euclidean_fitness_epsilon;          // registrator.cpp:901-904
    SlotGuard model, moving;
    d.check(mvr_cloud_transform(d.ctx(), model.s, raw_[0], model_->getPointCloud(object, 0).getMatrix().asColumnMajorColumnVector()), "mvr_cloud_transform");
    d.check(mvr_cloud_reserve(d.ctx(), model.s, total), "mvr_cloud_reserve");
    const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int v = 1; v < V; ++v) {
      ScanCloud &pc = model_->getPointCloud(object, v);
      pc.initRotation(*this);
      pc.setRegisterState(true);
      d.check(mvr_cloud_transform(d.ctx(), moving.s, raw_[(size_t)v], pc.getMatrix().asColumnMajorColumnVector()), "mvr_cloud_transform");
      for (int j = 0; j < repeat_times; ++j) {
        float T[16];
        mvr_icp_stats st = mvr_icp_stats();
        const int rc = mvr_icp_align(d.ctx(), moving.s, model.s, moving.s, &p, T, &st);            // out == in: the source advances in place
        if (rc != MVR_OK && rc != MVR_E_NOCORR) d.check(rc, "mvr_icp_align");
        AlignLog entry{pc.getView(), Matrix4f(T), st.n_corr, st.mse, st.iterations, 0.0, false};
        if (log_fitness) {
          d.check(mvr_fitness(d.ctx(), moving.s, model.s, identity, DBL_MAX, 0, &entry.fitness), "mvr_fitness");
          entry.has_fitness = true;
        }
        log.push_back(entry);
        pc.setMatrix(pc.getMatrix() * RowMatrixd(PclMatrixCaster<RowMatrixd>(entry.T)));
      }
      d.check(mvr_cloud_append(d.ctx(), model.s, moving.s), "mvr_cloud_append");
    }
  }

  // ---- global registration over the ring of views (what registrationLUM computes, registrator.cpp:611-664):
  // max(1, max_iterations / 16) outer passes; a pass = pose all views, reciprocal correspondences + sums of every ring
  // pair (i -> i+1), Lu-Milios relaxation (16 iterations) on the host from those sums, pose_v <- LUM_v o pose_v.
  // One native call (mvr_ring_run) runs ALL the passes.
  void registrationLUMDevice(int max_iterations, double max_distance, int object)
  {
    const int V = model_->numViews();
    if (V < 2) return;
    Device &d = Device::instance();
    ensureResident(object);
    std::vector<int> es((size_t)V), et((size_t)V);
    std::vector<double> poses((size_t)V * 16), rows((size_t)V * 32), pair_n((size_t)V), pair_mse((size_t)V), lum_pose((size_t)V * 6);
    for (int v = 0; v < V; ++v) {
      ScanCloud &pc = model_->getPointCloud(object, v);
      pc.initRotation(*this);
      pc.setRegisterState(true);
      std::memcpy(&poses[(size_t)v * 16], pc.getMatrix().asColumnMajorColumnVector(), 16 * sizeof(double));
      es[(size_t)v] = v; et[(size_t)v] = (v + 1) % V;
    }
    const int lum_iterations = 16, passes = std::max(1, max_iterations / lum_iterations);     // registrator.cpp:623-624
    int iters = 0;
    d.check(mvr_ring_run(d.ctx(), passes, V, posed_.data(), raw_.data(), V, es.data(), et.data(), max_distance, 1, 0, origin_,
                         lum_iterations, poses.data(), lum_pose.data(), nullptr, pair_n.data(), pair_mse.data(), &iters, rows.data(),
                         nullptr), "mvr_ring_run");
    lum_ncorr.clear();
    for (int e = 0; e < V; ++e) lum_ncorr.push_back((int)pair_n[(size_t)e]);
    for (int v = 0; v < V; ++v) {
      RowMatrixd m;
      std::memcpy(&m(0, 0), &poses[(size_t)v * 16], 16 * sizeof(double));
      model_->getPointCloud(object, v).setMatrix(m);
    }
  }

  // ---- residuals over the ring (what computeError gathers, registrator.cpp:466-515): neighbouring shown views
  // (i, i+1) plus (0, V-1); per pair the number of reciprocal correspondences within the threshold and the sum of
  // their squared distances.  All views are posed by one launch, all pairs searched by one launch per stage; only
  // 32 doubles per pair reach the host.
  struct PairResidual { int source, target; size_t n; double sum_d2; };
  std::vector<PairResidual> computeErrorDevice(int object, double distance_threshold)
  {
    const int V = model_->numViews();
    std::vector<PairResidual> out;
    if (V < 2) return out;
    Device &d = Device::instance();
    ensureResident(object);
    std::vector<char> shown((size_t)V, 0);
    std::vector<int> dst, src;
    std::vector<double> T;
    for (int v = 0; v < V; ++v) {
      ScanCloud &pc = model_->getPointCloud(object, v);
      shown[(size_t)v] = v == 0 || pc.isShown();
      if (!shown[(size_t)v]) continue;
      if (v) pc.initRotation(*this);
      dst.push_back(posed_[(size_t)v]); src.push_back(raw_[(size_t)v]);
      T.insert(T.end(), pc.getMatrix().asColumnMajorColumnVector(), pc.getMatrix().asColumnMajorColumnVector() + 16);
    }
    d.check(mvr_cloud_transform_batch(d.ctx(), (int)dst.size(), dst.data(), src.data(), T.data()), "mvr_cloud_transform_batch");
    std::vector<int> ps, pt;
    for (int i = 0; i + 1 < V; ++i) if (shown[(size_t)i] && shown[(size_t)i + 1]) { out.push_back(PairResidual{i, i + 1, 0, 0.0}); }
    if (V > 2 && shown[0] && shown[(size_t)V - 1]) out.push_back(PairResidual{0, V - 1, 0, 0.0});
    for (const PairResidual &r : out) { ps.push_back(posed_[(size_t)r.source]); pt.push_back(posed_[(size_t)r.target]); }
    if (out.empty()) return out;
    std::vector<mvr_pair_moments2_t> m2(out.size());
    d.check(mvr_pair_moments2_batch(d.ctx(), (int)out.size(), ps.data(), pt.data(), distance_threshold, 1, 0, nullptr, nullptr, origin_,
                                    m2.data(), nullptr), "mvr_pair_moments2_batch");
    for (size_t k = 0; k < out.size(); ++k) { out[k].n = (size_t)m2[k].n; out[k].sum_d2 = m2[k].sum_d2; }
    return out;
  }

  // ---- turntable axis and pivot from the registered views' poses (registrator.cpp:402-455): mvr_refine_axis.
  // Returns false (nothing changed) when no view but view 0 is registered or the poses do not determine an axis.
  bool refineAxis(int object)
  {
    const int V = model_->numViews();
    std::vector<double> poses;
    for (int v = 1; v < V; ++v) {
      const ScanCloud &pc = model_->getPointCloud(object, v);
      if (!pc.isRegistered()) continue;
      poses.insert(poses.end(), pc.getMatrix().asColumnMajorColumnVector(), pc.getMatrix().asColumnMajorColumnVector() + 16);
    }
    if (poses.empty()) return false;
    float axis[3], pivot[3];
    if (mvr_refine_axis((int)(poses.size() / 16), poses.data(), pivot_[1], axis, pivot) != MVR_OK) return false;
    for (int k = 0; k < 3; ++k) { axis_[k] = axis[k]; pivot_[k] = pivot[k]; }
    return true;
  }

  // ---- the merged object (registrator.cpp:344-400, the step registration() ends with, :719-744): every registered
  // view's points (and normals) moved by its pose -- the normals by the FULL affine including the translation, as
  // the reference does -- concatenated in view order and written as <folder>/points.pcd
  // (binary_compressed) and <folder>/points.asc.  Views that only hold XYZ points contribute zero colour / normals.
  // Returns the number of points written, 0 on failure.
  size_t saveRegisteredPoints(int object, const std::string &folder, io::RichCloud *merged_out = nullptr)
  {
    const int V = model_->numViews();
    io::RichCloud merged;
    for (int v = 0; v < V; ++v) {
      const ScanCloud &pc = model_->getPointCloud(object, v);
      if (!pc.isRegistered()) continue;
      const RowMatrixd &M = pc.getMatrix();
      auto pre = [&M](float x, float y, float z, float out[3]) {              // osg::Matrixd::preMult(Vec3f): (x, y, z, 1) * M / w
        const double w = 1.0 / (M(0, 3) * x + M(1, 3) * y + M(2, 3) * z + M(3, 3));
        for (int k = 0; k < 3; ++k) out[k] = (float)((M(0, k) * x + M(1, k) * y + M(2, k) * z + M(3, k)) * w);
      };
      for (size_t j = 0; j < pc.size(); ++j) {
        io::RichPoint q = j < pc.rich.size() ? pc.rich[j] : io::RichPoint();
        const PCLPoint &p = pc.points.points[j];
        float o[3];
        pre(p.x, p.y, p.z, o); q.x = o[0]; q.y = o[1]; q.z = o[2];
        pre(q.normal_x, q.normal_y, q.normal_z, o); q.normal_x = o[0]; q.normal_y = o[1]; q.normal_z = o[2];
        merged.push_back(q);
      }
    }
    const bool ok = io::savePCDFile(folder + "/points.pcd", merged, io::PCD_BINARY_COMPRESSED) && io::savePointsASC(folder + "/points.asc", merged);
    const size_t n = merged.size();
    if (merged_out) merged_out->swap(merged);
    return ok ? n : 0;
  }

  // ---- Registrator::registration (registrator.cpp:719-744): denoise every view, give it its prior, mark it
  // registered, save the merged cloud, refine the axis.
  size_t registration(int object, int segment_threshold, double triangle_length, const std::string &folder)
  {
    for (int v = 0; v < model_->numViews(); ++v) {
      ScanCloud &pc = model_->getPointCloud(object, v);
      pc.denoise(segment_threshold, triangle_length);
      pc.initRotation(*this);
      pc.setRegisterState(true);
    }
    const size_t n = saveRegisteredPoints(object, folder);
    refineAxis(object);
    return n;
  }

  // the scans of the model changed on the host (loaded, denoised, edited): upload them again at the next use
  void invalidate() { resident_object_ = -1; }

  std::vector<AlignLog> log;       // one entry per align (what the reference prints / writes to fitness_scores.txt)
  std::vector<int> lum_ncorr;      // correspondences per ring edge of the last global pass

 private:
  // raw scans -> device slots (once per object and per revision of its clouds); posed copies get slots of their own
  void ensureResident(int object)
  {
    const int V = model_->numViews();
    unsigned long long rev = 0;
    for (int v = 0; v < V; ++v) rev += model_->getPointCloud(object, v).revision() + (unsigned long long)model_->getPointCloud(object, v).size() * 1000003ull;
    if (resident_object_ == object && resident_rev_ == rev && (int)raw_.size() == V) return;
    Device &d = Device::instance();
    raw_guard_.clear(); posed_guard_.clear(); raw_.clear(); posed_.clear();
    for (int v = 0; v < V; ++v) {
      raw_guard_.emplace_back(new SlotGuard); posed_guard_.emplace_back(new SlotGuard);
      raw_.push_back(raw_guard_.back()->s); posed_.push_back(posed_guard_.back()->s);
      d.upload(raw_.back(), model_->getPointCloud(object, v).points);
    }
    origin_[0] = origin_[1] = origin_[2] = 0.0;           // moments are taken about a point of the data: small magnitudes
    if (V && !model_->getPointCloud(object, 0).points.empty()) {
      const PCLPoint &p = model_->getPointCloud(object, 0).points.points[0];
      origin_[0] = p.x; origin_[1] = p.y; origin_[2] = p.z;
    }
    resident_object_ = object; resident_rev_ = rev;
  }

  TurntableModel *model_;
  float pivot_[3] = {0, 0, 0};
  float axis_[3] = {0, 0, 1};
  int resident_object_ = -1;
  unsigned long long resident_rev_ = 0;
  std::vector<std::unique_ptr<SlotGuard> > raw_guard_, posed_guard_;
  std::vector<int> raw_, posed_;
  double origin_[3] = {0, 0, 0};
};

// point_cloud.cpp:400-413
inline void ScanCloud::initRotation(const Registrator &r)
{
  if (!getMatrix().isIdentity()) return;
  if (view == 0) return;
  setMatrix(r.getRotationMatrix(r.viewAngle(view)));
}

}  // namespace mvr
