// tests/cxx/asan_host_math.cpp -- driver of tests/test_host_asan.py: the host math (LUM loop, banded solve, Umeyama,
// one-call host step) on synthetic moments, built with -fsanitize=address,undefined
#include "../../multi-view-registration_amd/csrc/host_math.cpp"   // the product source itself, compiled for the host with the sanitizers on
#include <cstdio>
#include <random>
int main() {
  std::mt19937 g(7); std::normal_distribution<double> N(0, 1);
  for (int V : {2, 3, 5, 12, 36}) {
    const int ne = V;
    std::vector<int> es(ne), et(ne);
    for (int e = 0; e < ne; ++e) { es[e] = e; et[e] = (e + 1) % V; }
    std::vector<double> rows((size_t)ne * 32, 0.0);
    for (int e = 0; e < ne; ++e) {
      double *r = &rows[(size_t)e * 32];
      const int n = 3000; double t[3] = {0.02 * N(g), 0.02 * N(g), 0.02 * N(g)};
      for (int i = 0; i < n; ++i) {
        double p[3] = {60 * N(g), 30 * N(g), 10 * N(g)}, q[3];
        for (int k = 0; k < 3; ++k) q[k] = p[k] + t[k] + 0.05 * N(g);
        q[0] += 0.003 * p[1]; q[1] -= 0.003 * p[0];
        r[0] += 1; for (int k = 0; k < 3; ++k) { r[4 + k] += p[k]; r[7 + k] += q[k]; }
        int u = 0; for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b, ++u) { r[10 + u] += p[a] * p[b]; r[16 + u] += q[a] * q[b]; }
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) r[22 + 3 * a + b] += p[a] * q[b];
      }
    }
    if (V == 5) rows[32 * 2] = 1.0;        // an edge with too few correspondences
    std::vector<double> poses((size_t)V * 16, 0.0), lum((size_t)V * 6), pn(ne), pm(ne);
    for (int v = 0; v < V; ++v) for (int k = 0; k < 4; ++k) poses[(size_t)v * 16 + 5 * k] = 1.0;
    std::vector<float> pT((size_t)ne * 16);
    double origin[3] = {0, 0, 0}; int its = 0;
    const int rc = mvr_ring_host_step(V, ne, es.data(), et.data(), rows.data(), origin, 16, poses.data(), lum.data(), pT.data(), pn.data(), pm.data(), &its);
    printf("V=%d rc=%d its=%d lum[6]=%.6g pose[1][12]=%.6g\n", V, rc, its, lum[6], poses[16 + 12]);
  }
  // dense general graph (every pair an edge): wide rows, fill-in
  { const int V = 6; std::vector<int> es, et; for (int a = 0; a < V; ++a) for (int b = a + 1; b < V; ++b) { es.push_back(a); et.push_back(b); }
    const int ne = (int)es.size(); std::vector<mvr_pair_moments2_t> m2(ne);
    for (int e = 0; e < ne; ++e) { double *r = (double *)&m2[e]; std::memset(r, 0, sizeof m2[e]);
      for (int i = 0; i < 500; ++i) { double p[3] = {60 * N(g), 30 * N(g), 10 * N(g)}, q[3]; for (int k = 0; k < 3; ++k) q[k] = p[k] + 0.05 * N(g);
        r[0] += 1; for (int k = 0; k < 3; ++k) { r[4 + k] += p[k]; r[7 + k] += q[k]; }
        int u = 0; for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b, ++u) { r[10 + u] += p[a] * p[b]; r[16 + u] += q[a] * q[b]; }
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) r[22 + 3 * a + b] += p[a] * q[b]; } }
    std::vector<double> P((size_t)V * 6, 0.0); int its = 0;
    const int rc = mvr_lum_compute(V, ne, es.data(), et.data(), m2.data(), 5, 0.0, P.data(), &its);
    printf("complete graph rc=%d its=%d P[6]=%.6g\n", rc, its, P[6]); }
  return 0;
}
