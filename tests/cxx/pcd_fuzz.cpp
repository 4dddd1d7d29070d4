// tests/cxx/pcd_fuzz.cpp -- driver of tests/test_host_asan.py: the PCD reader (include/mvr/io.hpp) against damaged
// files.  Writes a small cloud in every PCD mode, then re-reads thousands of mutated copies (bytes flipped, sizes
// edited, truncated, spliced); built with -fsanitize=address,undefined, a reader that trusts a header field shows up
// as a sanitizer report.  The reader may accept or reject a damaged file -- it must not touch memory it does not own.
#include <cstdio>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "mvr/io.hpp"

int main(int argc, char **argv)
{
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  std::mt19937 g(12345);
  mvr::io::RichCloud cloud;
  for (int i = 0; i < 700; ++i) {
    mvr::io::RichPoint p{};
    p.x = (float)(i % 31) * 0.5f; p.y = (float)(i % 7); p.z = 900.0f + (float)(i % 3);      // compressible
    p.normal_x = 0.f; p.normal_y = 0.f; p.normal_z = 1.f;
    p.r = (uint8_t)(i & 255); p.g = 10; p.b = 200;
    cloud.push_back(p);
  }
  size_t accepted = 0, rejected = 0;
  for (int mode = 0; mode < 3; ++mode) {
    const std::string f = dir + "/m" + std::to_string(mode) + ".pcd";
    if (!mvr::io::savePCDFile(f, cloud, (mvr::io::PcdMode)mode)) return 3;
    std::ifstream in(f, std::ios::binary);
    const std::vector<char> good((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    for (int trial = 0; trial < 1500; ++trial) {
      std::vector<char> bad = good;
      const int kind = trial % 5;
      if (kind == 0) { for (int k = 0; k < 1 + (int)(g() % 8); ++k) bad[g() % bad.size()] ^= (char)(1u << (g() % 8)); }
      else if (kind == 1) bad.resize(g() % bad.size());                                         // truncated
      else if (kind == 2) { const size_t at = g() % std::min<size_t>(bad.size(), 400); bad[at] = (char)('0' + g() % 10); }   // header digits
      else if (kind == 3) { const size_t a = g() % bad.size(), n = g() % 64; bad.insert(bad.begin() + a, n, (char)(g() & 255)); }
      else { const size_t a = g() % bad.size(), b = g() % bad.size(); for (size_t k = 0; k < 32 && a + k < bad.size() && b + k < bad.size(); ++k) bad[a + k] = good[b + k]; }
      const std::string fb = dir + "/bad.pcd";
      { std::ofstream out(fb, std::ios::binary); out.write(bad.data(), (std::streamsize)bad.size()); }
      mvr::io::RichCloud c2;
      if (mvr::io::loadPCDFile(fb, c2)) ++accepted; else ++rejected;
    }
  }
  // headers that promise far more than the file holds: rejected before anything is allocated for them
  for (const char *data : {"ascii", "binary", "binary_compressed"})
    for (const char *pts : {"99999999999", "4294967295", "18446744073709551615"}) {
      const std::string fb = dir + "/huge.pcd";
      { std::ofstream out(fb, std::ios::binary);
        out << "VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH " << pts << "\nHEIGHT 1\nPOINTS " << pts << "\nDATA " << data << "\n";
        const unsigned int lie[2] = {16u, 0xFFFFFFF0u}; out.write((const char *)lie, 8); out << "0 0 0\n1 1 1\n"; }
      mvr::io::RichCloud c2;
      if (mvr::io::loadPCDFile(fb, c2)) { std::printf("accepted a header of %s points (%s)\n", pts, data); return 4; }
      ++rejected;
    }
  { const std::string fb = dir + "/count.pcd";          // a field count that would overflow the point stride
    std::ofstream out(fb, std::ios::binary);
    out << "FIELDS x y z\nSIZE 4 4 8\nTYPE F F F\nCOUNT 1 1 2000000000\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA binary\n0123456789ab";
    out.close();
    mvr::io::RichCloud c2;
    if (mvr::io::loadPCDFile(fb, c2)) return 5;
    ++rejected; }
  std::printf("accepted=%zu rejected=%zu\n", accepted, rejected);
  return 0;
}
