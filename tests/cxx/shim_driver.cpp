// tests/cxx/shim_driver.cpp -- drives the C++ drop-in shim (include/mvr/*.hpp)
// exactly the way the reference's Registrator drives PCL, on synthetic
// turntable scans, and prints the results as JSON for tests/test_gpu_shim.py
// to compare with the oracle-based restatement of the same driver loops.
//
// usage: shim_driver <mode> <views> <points> <max_dist> <repeat> <config_id>
//   mode: seq | lum | auto | err | api          the PCL-named call surface, in the reference's order of calls (call_surface.hpp)
//         seqdev | lumdev | autodev | errdev | register   the product's device-resident drivers (mvr/registrator.hpp)
//         world                                    the single-process multi-GPU host (mvr_world_*), here with one GPU
//         denoise
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>

#define MVR_ALIAS_PCL
#include "call_surface.hpp"

using namespace mvr;
using surface::CallSurface;

static void print_pose(const RowMatrixd &m, bool last)
{
  // column-vector 4x4, row by row (T(r,c) = m(c,r)) -- the transformation.txt order (point_cloud.cpp:336-343)
  std::printf("[");
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) std::printf("%.17g%s", m(c, r), (r == 3 && c == 3) ? "" : ",");
  std::printf("]%s", last ? "" : ",");
}

static void print_mat4f(const Matrix4f &T)
{
  std::printf("[");
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) std::printf("%.9g%s", T(r, c), (r == 3 && c == 3) ? "" : ",");
  std::printf("]");
}

int main(int argc, char **argv)
{
  if (argc < 7) { std::fprintf(stderr, "usage: %s mode views points max_dist repeat config\n", argv[0]); return 2; }
  const std::string mode = argv[1];
  const int V = std::atoi(argv[2]);
  const size_t N = (size_t)std::atol(argv[3]);
  const double max_d = std::atof(argv[4]);
  const int repeat = std::atoi(argv[5]);
  const int config = std::atoi(argv[6]);
  try {
    mvr_synth_params sp;
    mvr_synth_default(&sp, V, config);
    TurntableModel model;
    model.views.resize(V);
    for (int v = 0; v < V; ++v) {
      model.views[v].view = v;
      model.views[v].points.resize(N);
      if (mvr_synth_view(&sp, v, N, model.views[v].points.points[0].data, nullptr) != MVR_OK) return 3;
    }
    CallSurface reg(&model);             // is-a mvr::Registrator: the device-resident drivers are reachable through it too
    double piv[3], ax[3];
    mvr_synth_prior(&sp, piv, ax);
    reg.setPivotPoint(piv[0], piv[1], piv[2]);
    reg.setAxisNormal(ax[0], ax[1], ax[2]);

    std::printf("{\"mode\":\"%s\",", mode.c_str());
    if (mode == "seq") {
      surface::IcpKnobs k;                       // the sequential mode's settings: registrator.cpp:551-560
      k.max_distance = max_d; k.max_iterations = 1000; k.transformation_epsilon = 0.000001; k.fitness_epsilon = 64;
      reg.growingTargetSweeps(k, repeat);
    } else if (mode == "lum") {
      reg.ringRelaxation(repeat, max_d);
      std::printf("\"lum_ncorr\":[");
      for (size_t i = 0; i < reg.lum_ncorr.size(); ++i) std::printf("%d%s", reg.lum_ncorr[i], i + 1 < reg.lum_ncorr.size() ? "," : "");
      std::printf("],");
    } else if (mode == "seqdev") {
      reg.registrationICPDevice(1000, max_d, 0, repeat);
    } else if (mode == "autodev") {
      reg.automaticRegistrationDevice(0, 1000, repeat, max_d, 50.0, true);
    } else if (mode == "errdev") {
      auto pairs = reg.computeErrorDevice(0, max_d);
      std::printf("\"pairs\":[");
      for (size_t i = 0; i < pairs.size(); ++i)
        std::printf("[%d,%d,%zu,%.17g]%s", pairs[i].source, pairs[i].target, pairs[i].n, pairs[i].sum_d2, i + 1 < pairs.size() ? "," : "");
      std::printf("],");
    } else if (mode == "register") {
      // Registrator::registration: denoise + prior + merged cloud on disk (points.pcd / points.asc) + refined axis;
      // argv[7] = output folder
      const std::string folder = argc > 7 ? argv[7] : ".";
      for (int v = 0; v < V; ++v) {       // colours / normals ride along: view v is painted (v, 2v, 255 - v)
        ScanCloud &pc = model.views[v];
        pc.rich.resize(pc.points.size());
        for (size_t i = 0; i < pc.points.size(); ++i) {
          io::RichPoint &q = pc.rich[i];
          q.x = pc.points.points[i].x; q.y = pc.points.points[i].y; q.z = pc.points.points[i].z;
          q.r = (uint8_t)v; q.g = (uint8_t)(2 * v); q.b = (uint8_t)(255 - v); q.normal_x = 0.f; q.normal_y = 0.f; q.normal_z = 1.f;
        }
      }
      const size_t n = reg.registration(0, 10, 2.5, folder);
      io::RichCloud back;
      const bool ok = io::loadPCDFile(folder + "/points.pcd", back);
      std::printf("\"merged\":%zu,\"reloaded\":%zu,\"ok\":%d,\"sizes\":[", n, back.size(), (int)ok);
      for (int v = 0; v < V; ++v) std::printf("%zu%s", model.views[v].size(), v + 1 < V ? "," : "");
      std::printf("],");
    } else if (mode == "world") {
      // the native multi-GPU host, one process: mvr_world_create(1) (ncclCommInitAll against the SYSTEM RCCL: no torch in
      // this process) + mvr_world_ring_run == registrationLUMDevice on a plain context
      mvr_world *w = nullptr;
      std::fflush(stdout);                       // RCCL prints a version banner on fd 1: keep it out of the JSON
      const int saved_stdout = dup(1);
      dup2(2, 1);
      const int rc = mvr_world_create(&w, 1, nullptr);
      if (rc != MVR_OK) { std::fprintf(stderr, "mvr_world_create: %s (%s)\n", mvr_strerror(rc), mvr_rccl_library()); return 1; }
      std::vector<int> raw_s((size_t)V), posed_s((size_t)V), es((size_t)V), et((size_t)V);
      std::vector<double> poses((size_t)V * 16), lum((size_t)V * 6), pn((size_t)V), pm((size_t)V);
      for (int v = 0; v < V; ++v) {
        ScanCloud &pc = model.views[v];
        pc.initRotation(reg);
        raw_s[v] = V + v; posed_s[v] = v; es[v] = v; et[v] = (v + 1) % V;
        if (mvr_world_upload(w, V + v, pc.points.points[0].data, pc.size(), 16) != MVR_OK) return 1;
        std::memcpy(&poses[(size_t)v * 16], pc.getMatrix().asColumnMajorColumnVector(), 16 * sizeof(double));
      }
      const PCLPoint &p0 = model.views[0].points.points[0];
      const double origin[3] = {p0.x, p0.y, p0.z};
      int iters = 0;
      const int rr = mvr_world_ring_run(w, repeat, V, posed_s.data(), raw_s.data(), V, es.data(), et.data(), max_d, 1, 0, origin, 16, poses.data(),
                                        lum.data(), nullptr, pn.data(), pm.data(), &iters, nullptr, nullptr);
      if (rr != MVR_OK) { std::fprintf(stderr, "mvr_world_ring_run: %s (%s)\n", mvr_strerror(rr), mvr_world_last_error(w)); return 1; }
      int rank = -1, world = -1, rccl = -1;
      mvr_ctx_comm_info(mvr_world_ctx(w, 0), &rank, &world, &rccl);
      std::fflush(stdout);
      dup2(saved_stdout, 1);
      close(saved_stdout);
      std::printf("\"rccl\":\"%s\",\"comm\":[%d,%d,%d],\"lum_ncorr\":[", mvr_rccl_library(), rank, world, rccl);
      for (int e = 0; e < V; ++e) std::printf("%d%s", (int)pn[e], e + 1 < V ? "," : "");
      std::printf("],");
      for (int v = 0; v < V; ++v) { RowMatrixd m; std::memcpy(&m(0, 0), &poses[(size_t)v * 16], 16 * sizeof(double)); model.views[v].setMatrix(m); }
      mvr_world_destroy(w);
    } else if (mode == "lumdev") {
      reg.registrationLUMDevice(16 * repeat, max_d, 0);
      std::printf("\"lum_ncorr\":[");
      for (size_t i = 0; i < reg.lum_ncorr.size(); ++i) std::printf("%d%s", reg.lum_ncorr[i], i + 1 < reg.lum_ncorr.size() ? "," : "");
      std::printf("],");
    } else if (mode == "denoise") {
      // scan 0 plus far-away outliers (one per 40 points), then PointCloud::denoise(10, 2.5)
      ScanCloud &pc = model.views[0];
      const size_t n0 = pc.points.size();
      for (size_t k = 0; k < n0 / 40; ++k) pc.points.push_back(PointXYZ(1000.f + 7.f * (float)k, -500.f, 2000.f + 3.f * (float)(k % 11)));
      const size_t noise = pc.denoise(10, 2.5);
      double sx = 0, sy = 0, sz = 0;
      for (size_t i = 0; i < pc.points.size(); ++i) { sx += pc.points.points[i].x; sy += pc.points.points[i].y; sz += pc.points.points[i].z; }
      std::printf("\"noise\":%zu,\"kept\":%zu,\"sum\":[%.17g,%.17g,%.17g],", noise, pc.points.size(), sx, sy, sz);
    } else if (mode == "auto") {
      surface::IcpKnobs k;                       // the incremental mode: no transformation epsilon (registrator.cpp:901-904), fitness epsilon 50
      k.max_distance = max_d; k.max_iterations = 1000; k.with_transformation_epsilon = false; k.fitness_epsilon = 50.0;
      reg.addViewsOneByOne(k, repeat);
    } else if (mode == "err") {
      auto pairs = reg.ringResiduals(max_d);
      std::printf("\"pairs\":[");
      for (size_t i = 0; i < pairs.size(); ++i) {
        double s = 0; for (const Correspondence &c : *pairs[i].pairs) s += c.distance;
        std::printf("[%d,%d,%zu,%.17g]%s", pairs[i].a, pairs[i].b, pairs[i].pairs->size(), s, i + 1 < pairs.size() ? "," : "");
      }
      std::printf("],");
    } else if (mode == "api") {
      // the PCL-named API surface, through the pcl:: alias, incl. the error paths
      pcl::PointCloud<pcl::PointXYZ>::Ptr a(new pcl::PointCloud<pcl::PointXYZ>), b(new pcl::PointCloud<pcl::PointXYZ>);
      model.views[0].getTransformedPoints(*a);
      model.views[1].initRotation(reg);
      model.views[1].getTransformedPoints(*b);
      pcl::IterativeClosestPoint<pcl::PointXYZ, pcl::PointXYZ> icp;
      icp.setUseReciprocalCorrespondences(true);
      icp.setMaxCorrespondenceDistance(max_d);
      icp.setMaximumIterations(7);
      icp.setTransformationEpsilon(0);
      icp.setEuclideanFitnessEpsilon(-1e300);
      icp.setInputSource(b);
      icp.setInputTarget(a);
      icp.align(*b);                                   // aliased output (registrator.cpp:920)
      std::printf("\"iters\":%d,\"converged\":%d,\"T\":", icp.getStats().iterations, (int)icp.hasConverged());
      print_mat4f(icp.getFinalTransformation());
      std::printf(",\"fitness_after_alias\":%.17g,", icp.getFitnessScore());
      // far-away target: not enough correspondences -> no throw, hasConverged() false
      pcl::PointCloud<pcl::PointXYZ>::Ptr far(new pcl::PointCloud<pcl::PointXYZ>(*a));
      for (auto &p : far->points) p.x += 1e4f;
      icp.setMaxCorrespondenceDistance(1.0);
      icp.setInputTarget(far);
      pcl::PointCloud<pcl::PointXYZ> out;
      icp.align(out);
      std::printf("\"nocorr_converged\":%d,\"nocorr_identity\":%d,", (int)icp.hasConverged(), (int)icp.getFinalTransformation().isIdentity());
      pcl::registration::CorrespondenceEstimation<pcl::PointXYZ, pcl::PointXYZ, float> ce;
      ce.setInputSource(b); ce.setInputTarget(a);
      pcl::Correspondences one, rec;
      ce.determineCorrespondences(one, max_d);
      ce.determineReciprocalCorrespondences(rec, max_d);
      std::printf("\"n_oneway\":%zu,\"n_recip\":%zu,", one.size(), rec.size());
    } else {
      std::fprintf(stderr, "unknown mode\n");
      return 2;
    }
    std::printf("\"log\":[");
    for (size_t i = 0; i < reg.log.size(); ++i) {
      const AlignLog &e = reg.log[i];
      std::printf("{\"view\":%d,\"n_corr\":%d,\"mse\":%.17g,\"iterations\":%d,\"fitness\":%s,\"T\":", e.view, e.n_corr, e.mse, e.iterations,
                  e.has_fitness ? std::to_string(e.fitness).c_str() : "null");
      print_mat4f(e.T);
      std::printf("}%s", i + 1 < reg.log.size() ? "," : "");
    }
    std::printf("],");
    if (mode == "seq" || mode == "auto" || mode == "autodev" || mode == "lum" || mode == "lumdev" || mode == "seqdev") {
      for (int v = 0; v < V; ++v) model.views[v].setRegisterState(true);
      reg.refineAxis(0);                          // the product's (mvr_refine_axis)
      std::printf("\"refined_pivot\":[%.9g,%.9g,%.9g],\"refined_axis\":[%.9g,%.9g,%.9g],", reg.getPivotPoint()[0], reg.getPivotPoint()[1],
                  reg.getPivotPoint()[2], reg.getAxisNormal()[0], reg.getAxisNormal()[1], reg.getAxisNormal()[2]);
    }
    std::printf("\"poses\":[");
    for (int v = 0; v < V; ++v) print_pose(model.views[v].getMatrix(), v == V - 1);
    std::printf("]}\n");
  } catch (const mvr::Error &e) {
    std::fprintf(stderr, "mvr::Error %d: %s\n", e.status, e.what());
    return 1;
  }
  return 0;
}
