// tests/cxx/host_driver.cpp -- host-only parts of the C++ shim, runnable without
// a GPU: the reference's text formats (transformation.txt, axis.txt), the
// OSG <-> Eigen matrix bridge (PclMatrixCaster), the turntable prior and
// Registrator::refineAxis.  Prints JSON for tests/test_shim_host.py.
#include <cstdio>
#include <string>

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
  std::printf("\"refined\":[%.9g,%.9g,%.9g,%.9g,%.9g,%.9g]", reg3.getPivotPoint()[0], reg3.getPivotPoint()[1], reg3.getPivotPoint()[2],
              reg3.getAxisNormal()[0], reg3.getAxisNormal()[1], reg3.getAxisNormal()[2]);
  std::printf("}\n");
  return 0;
}
