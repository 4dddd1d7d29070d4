// tests/cxx/host_driver.cpp -- host-only parts of the C++ shim, runnable without
// a GPU: the reference's text formats (transformation.txt, axis.txt), the
// OSG <-> Eigen matrix bridge (PclMatrixCaster), the turntable prior and
// Registrator::refineAxis.  Prints JSON for tests/test_shim_host.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "mvr/io.hpp"
#include "mvr/registrator.hpp"

using namespace mvr;

static void print16(const char *name, const RowMatrixd &m, bool last = false)
{
  std::printf("\"%s\":[", name);
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) std::printf("%.17g%s", m(r, c), (r == 3 && c == 3) ? "" : ",");
  std::printf("]%s", last ? "" : ",");
}

int main(int argc, char **argv)
{
  const std::string dir = argc > 1 ? argv[1] : ".";
  TurntableModel model;
  const int V = 12;
  model.views.resize(V);
  for (int v = 0; v < V; ++v) model.views[v].view = v;
  Registrator reg(&model);
  // the true turntable (mvr/src/point_cloud.cpp:102-103)
  const double piv[3] = {-13.382786, 50.223461, 917.4776}, ax[3] = {-0.054323, -0.814921, -0.577020};
  reg.setPivotPoint(piv[0], piv[1], piv[2]);
  reg.setAxisNormal(ax[0], ax[1], ax[2]);

  std::printf("{");
  // initRotation: identity stays for view 0, others get the prior; a non-identity pose is kept
  for (int v = 0; v < V; ++v) model.views[v].initRotation(reg);
  print16("prior_view0", model.views[0].getMatrix());
  print16("prior_view1", model.views[1].getMatrix());
  print16("prior_view7", model.views[7].getMatrix());
  RowMatrixd keep = model.views[3].getMatrix();
  model.views[3].initRotation(reg);
  std::printf("\"init_keeps_pose\":%d,", (int)(std::memcmp(&keep, &model.views[3].getMatrix(), sizeof keep) == 0));

  // transformation.txt round trip (6 decimals) and exact file text
  const std::string tf = dir + "/transformation.txt";
  std::printf("\"save_tf\":%d,", (int)model.views[1].saveTransformation(tf));
  ScanCloud loaded;
  std::printf("\"load_tf\":%d,", (int)loaded.loadTransformation(tf));
  print16("loaded_view1", loaded.getMatrix());
  std::printf("\"load_missing\":%d,", (int)loaded.loadTransformation(dir + "/does_not_exist.txt"));

  // axis.txt
  const std::string af = dir + "/axis.txt";
  std::printf("\"save_axis\":%d,", (int)reg.save(af));
  Registrator reg2(&model);
  std::printf("\"load_axis\":%d,", (int)reg2.load(af));
  std::printf("\"axis_loaded\":[%.9g,%.9g,%.9g,%.9g,%.9g,%.9g],", reg2.getPivotPoint()[0], reg2.getPivotPoint()[1], reg2.getPivotPoint()[2],
              reg2.getAxisNormal()[0], reg2.getAxisNormal()[1], reg2.getAxisNormal()[2]);

  // PclMatrixCaster: Eigen (column-vector) <-> OSG (row-vector) is a transpose both ways
  Matrix4f e = PclMatrixCaster<RowMatrixd>(model.views[1].getMatrix());
  RowMatrixd back = PclMatrixCaster<RowMatrixd>(e);
  std::printf("\"caster_e\":[");
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) std::printf("%.9g%s", e(r, c), (r == 3 && c == 3) ? "" : ",");
  std::printf("],");
  print16("caster_back", back);

  // refineAxis: poses that are exact rotations about the TRUE (pivot, axis) but a
  // mis-calibrated starting estimate -> the least-squares fit returns to the truth
  Registrator reg3(&model);
  reg3.setPivotPoint(piv[0] + 1.5, piv[1] - 1.0, piv[2] + 2.0);
  reg3.setAxisNormal(ax[0], ax[1] * 0.99, ax[2] * 1.02);
  for (int v = 0; v < V; ++v) model.views[v].setRegisterState(true);   // poses from `reg` above = exact
  reg3.refineAxis(0);
  std::printf("\"refined\":[%.9g,%.9g,%.9g,%.9g,%.9g,%.9g],", reg3.getPivotPoint()[0], reg3.getPivotPoint()[1], reg3.getPivotPoint()[2],
              reg3.getAxisNormal()[0], reg3.getAxisNormal()[1], reg3.getAxisNormal()[2]);

  // ---- PCD / points.asc (mvr/io.hpp): a deterministic rich cloud through all three encodings and back
  io::RichCloud cloud(1037);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)(st >> 8) / 16777216.0f; };
  for (size_t i = 0; i < cloud.size(); ++i) {
    io::RichPoint &p = cloud[i];
    p.x = 200.f * rnd() - 100.f; p.y = 200.f * rnd() - 100.f; p.z = 900.f + 50.f * rnd();
    p.r = (uint8_t)(255 * rnd()); p.g = (uint8_t)(i & 255); p.b = (uint8_t)((i * 7) & 255);
    p.normal_x = rnd(); p.normal_y = rnd(); p.normal_z = rnd(); p.curvature = 0.01f * rnd();
    if (i >= 500) { p.normal_x = 0.f; p.normal_y = 0.f; p.normal_z = 1.f; p.curvature = 0.f; }   // long runs: LZF back references
  }
  const char *names[3] = {"ascii", "binary", "compressed"};
  std::printf("\"pcd\":{");
  for (int m = 0; m < 3; ++m) {
    const std::string fn = dir + "/cloud_" + names[m] + ".pcd";
    const bool saved = io::savePCDFile(fn, cloud, (io::PcdMode)m);
    io::RichCloud back;
    const bool loaded = io::loadPCDFile(fn, back);
    bool same = loaded && back.size() == cloud.size();
    for (size_t i = 0; same && i < cloud.size(); ++i)
      same = std::memcmp(&back[i].x, &cloud[i].x, 12) == 0 && back[i].r == cloud[i].r && back[i].g == cloud[i].g && back[i].b == cloud[i].b &&
             back[i].normal_x == cloud[i].normal_x && back[i].normal_y == cloud[i].normal_y && back[i].normal_z == cloud[i].normal_z &&
             back[i].curvature == cloud[i].curvature;
    std::printf("\"%s\":[%d,%d,%d],", names[m], (int)saved, (int)loaded, (int)same);
  }
  {
    // a file another writer could have produced: different field order, f64 coordinates, u8 extra field, CRLF header
    const std::string fn = dir + "/foreign.pcd";
    FILE *f = std::fopen(fn.c_str(), "wb");
    std::fprintf(f, "# foreign\r\nVERSION .7\r\nFIELDS label z y x\r\nSIZE 1 8 8 8\r\nTYPE U F F F\r\nCOUNT 1 1 1 1\r\nWIDTH 3\r\nHEIGHT 1\r\nPOINTS 3\r\nDATA binary\n");
    for (int i = 0; i < 3; ++i) { uint8_t l = (uint8_t)i; double v[3] = {30.0 + i, 20.0 + i, 10.0 + i}; std::fwrite(&l, 1, 1, f); std::fwrite(v, 8, 3, f); }
    std::fclose(f);
    io::RichCloud back;
    const bool ok = io::loadPCDFile(fn, back);
    std::printf("\"foreign\":[%d,%zu,%.9g,%.9g,%.9g],", (int)ok, back.size(), ok ? back[2].x : 0.f, ok ? back[2].y : 0.f, ok ? back[2].z : 0.f);
    // truncated payload / no xyz / missing file -> false, cloud untouched
    io::RichCloud keep(2);
    const std::string tr = dir + "/truncated.pcd";
    { std::ifstream in((dir + "/cloud_compressed.pcd").c_str(), std::ios::binary); std::string all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
      std::ofstream o(tr.c_str(), std::ios::binary); o.write(all.data(), (std::streamsize)(all.size() * 2 / 3)); }
    const bool t1 = io::loadPCDFile(tr, keep);
    const std::string nx = dir + "/noxyz.pcd";
    { FILE *g = std::fopen(nx.c_str(), "w"); std::fprintf(g, "VERSION .7\nFIELDS a b\nSIZE 4 4\nTYPE F F\nCOUNT 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2\n"); std::fclose(g); }
    const bool t2 = io::loadPCDFile(nx, keep), t3 = io::loadPCDFile(dir + "/nope.pcd", keep);
    std::printf("\"rejects\":[%d,%d,%d,%zu],", (int)t1, (int)t2, (int)t3, keep.size());
  }
  std::printf("\"asc\":%d,", (int)io::savePointsASC(dir + "/points.asc", cloud));
  std::printf("\"path\":\"%s\"},", io::pointsFilename("/data/ws", 7, 3).c_str());
  PointCloud<PointXYZ> xyz;
  io::toXYZ(cloud, xyz);
  std::printf("\"xyz\":[%zu,%.9g,%.9g],", xyz.size(), xyz.points[5].x, xyz.points[5].data[3]);

  // ---- ScanCloud::open / save (point_cloud.cpp:78-123): PCD + the transformation.txt beside it; "*.ply" writes the
  // XYZ points in the turntable's canonical frame (pivot -> origin, axis -> +z)
  {
    ScanCloud sc; sc.view = 1;
    const bool missing = sc.open(dir + "/nope.pcd");
    io::RichCloud three(3);
    const double a = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    three[0].x = (float)piv[0]; three[0].y = (float)piv[1]; three[0].z = (float)piv[2];
    three[1].x = (float)(piv[0] + 10.0 * ax[0] / a); three[1].y = (float)(piv[1] + 10.0 * ax[1] / a); three[1].z = (float)(piv[2] + 10.0 * ax[2] / a);
    three[2].x = (float)(piv[0] + 5.0); three[2].y = (float)piv[1]; three[2].z = (float)piv[2]; three[2].r = 9;
    io::savePCDFile(dir + "/points.pcd", three, io::PCD_BINARY_COMPRESSED);       // dir also holds view 1's transformation.txt (saved above)
    const bool opened = sc.open(dir + "/points.pcd");
    std::printf("\"open\":[%d,%d,%zu,%zu,%d,%d],", (int)missing, (int)opened, sc.size(), sc.rich.size(), (int)sc.isRegistered(), (int)sc.rich[2].r);
    print16("opened_pose", sc.getMatrix());
    const bool ply = sc.save(dir + "/canon.ply"), pcd = sc.save(dir + "/copy.pcd");
    PointCloud<PointXYZ> canon;
    const bool plyback = io::loadPLYFile(dir + "/canon.ply", canon);
    io::RichCloud copy;
    const bool pcdback = io::loadPCDFile(dir + "/copy.pcd", copy);
    std::printf("\"ply\":[%d,%d,%d,%d,%zu,%zu,%d],\"canon\":[", (int)ply, (int)pcd, (int)plyback, (int)pcdback, canon.size(), copy.size(), (int)copy[2].r);
    for (size_t i = 0; i < canon.size(); ++i) std::printf("%.9g,%.9g,%.9g%s", canon.points[i].x, canon.points[i].y, canon.points[i].z, i + 1 < canon.size() ? "," : "");
    std::printf("],");
    // a PLY another writer could have produced: double coordinates, an extra property, a face element
    { FILE *f = std::fopen((dir + "/foreign.ply").c_str(), "w");
      std::fprintf(f, "ply\nformat ascii 1.0\nelement vertex 2\nproperty double z\nproperty uchar q\nproperty double x\nproperty double y\nelement face 0\nproperty int n\nend_header\n3 7 1 2\n6 8 4 5\n");
      std::fclose(f); }
    PointCloud<PointXYZ> fp;
    const bool fok = io::loadPLYFile(dir + "/foreign.ply", fp);
    { FILE *f = std::fopen((dir + "/binary.ply").c_str(), "w"); std::fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nend_header\n"); std::fclose(f); }
    PointCloud<PointXYZ> keep; keep.push_back(PointXYZ(1, 2, 3));
    const bool bok = io::loadPLYFile(dir + "/binary.ply", keep);
    std::printf("\"foreign_ply\":[%d,%zu,%.9g,%.9g,%.9g,%d,%zu],", (int)fok, fp.size(), fok ? fp.points[1].x : 0.f, fok ? fp.points[1].y : 0.f, fok ? fp.points[1].z : 0.f, (int)bok, keep.size());
  }
  // ---- Registrator::saveRegisteredPoints (registrator.cpp:344-400): registered views only, points AND normals moved
  // by the full pose (translation included, App. C.5), colours kept; points.pcd + points.asc
  {
    TurntableModel m2; m2.views.resize(3);
    for (int v = 0; v < 3; ++v) {
      ScanCloud &pc = m2.views[v]; pc.view = v;
      for (int i = 0; i < 4 + v; ++i) {
        pc.points.push_back(PointXYZ(10.f * v + i, -3.f + i, 900.f + v));
        io::RichPoint q; q.x = pc.points.points[i].x; q.y = pc.points.points[i].y; q.z = pc.points.points[i].z;
        q.r = (uint8_t)(40 + v); q.g = (uint8_t)i; q.b = 7; q.normal_x = 0.f; q.normal_y = 0.6f; q.normal_z = 0.8f;
        pc.rich.push_back(q);
      }
    }
    m2.views[1].setMatrix(model.views[1].getMatrix());
    m2.views[0].setRegisterState(true); m2.views[1].setRegisterState(true);      // view 2 is NOT registered: left out
    Registrator r2(&m2);
    const std::string out_dir = dir + "/merged";
    std::string cmd = "mkdir -p '" + out_dir + "'";
    if (std::system(cmd.c_str()) != 0) return 9;
    io::RichCloud merged;
    const size_t n = r2.saveRegisteredPoints(0, out_dir, &merged);
    io::RichCloud back;
    const bool ok = io::loadPCDFile(out_dir + "/points.pcd", back);
    std::printf("\"merged\":[%zu,%d,%zu,%d,%d],", n, (int)ok, back.size(), (int)back[4].r, (int)back[4].g);
    std::printf("\"merged_p4\":[%.9g,%.9g,%.9g,%.9g,%.9g,%.9g],", back[4].x, back[4].y, back[4].z, back[4].normal_x, back[4].normal_y, back[4].normal_z);
    std::printf("\"merged_p0\":[%.9g,%.9g,%.9g,%.9g,%.9g,%.9g]", back[0].x, back[0].y, back[0].z, back[0].normal_x, back[0].normal_y, back[0].normal_z);
  }
  std::printf("}\n");
  return 0;
}
