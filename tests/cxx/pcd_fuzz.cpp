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
  // SIZE / TYPE cross-mutations: every combination of the header's sizes and types on a body that is
  // exactly points * step bytes long -- a decoder that trusts TYPE over SIZE reads past the end of the last record
  // (ASan reports it); float fields of 1 or 2 bytes and a colour that is not 4 bytes wide must be refused.
  {
    const int sizes[4] = {1, 2, 4, 8};
    const char types[3] = {'F', 'U', 'I'};
    size_t combos = 0;
    for (const char *data : {"binary", "binary_compressed"})
      for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) for (int c = 0; c < 12; ++c) for (int last = 0; last < 2; ++last) {
        const int sz[3] = {sizes[a % 4], sizes[b % 4], sizes[c % 4]};
        const char ty[3] = {types[a / 4], types[b / 4], types[c / 4]};
        const char *third = last ? "rgb" : "z";           // the colour field is decoded as 4 bytes whatever TYPE says
        const std::string fields = last ? std::string("x y z rgb") : std::string("x y z");
        const size_t step = (size_t)sz[0] + sz[1] + (last ? 4 + (size_t)sz[2] : (size_t)sz[2]);
        const size_t npts = 3;
        std::vector<uint8_t> body(npts * step, 0x3f);
        const std::string fb = dir + "/cross.pcd";
        { std::ofstream out(fb, std::ios::binary);
          out << "VERSION .7\nFIELDS " << fields << "\nSIZE " << sz[0] << " " << sz[1] << " " << (last ? 4 : sz[2]);
          if (last) out << " " << sz[2];
          out << "\nTYPE " << ty[0] << " " << ty[1] << " " << (last ? 'F' : ty[2]);
          if (last) out << " " << ty[2];
          out << "\nCOUNT 1 1 1" << (last ? " 1" : "") << "\nWIDTH " << npts << "\nHEIGHT 1\nPOINTS " << npts << "\nDATA " << data << "\n";
          if (std::string(data) == "binary") out.write((const char *)body.data(), (std::streamsize)body.size());
          else {
            std::vector<uint8_t> comp;
            mvr::io::lzf_compress(body.data(), body.size(), comp);
            const uint32_t cs = (uint32_t)comp.size(), us = (uint32_t)body.size();
            out.write((const char *)&cs, 4); out.write((const char *)&us, 4); out.write((const char *)comp.data(), (std::streamsize)comp.size());
          } }
        (void)third;
        mvr::io::RichCloud c2;
        const bool ok = mvr::io::loadPCDFile(fb, c2);
        bool must_reject = false;
        for (int k = 0; k < 3; ++k) must_reject |= (ty[k] == 'F' && sz[k] != 4 && sz[k] != 8);
        if (last && sz[2] != 4) must_reject = true;
        if (ok && must_reject) { std::printf("accepted SIZE %d %d %d TYPE %c %c %c (%s)\n", sz[0], sz[1], sz[2], ty[0], ty[1], ty[2], data); return 6; }
        if (!ok && !must_reject) { std::printf("refused a consistent header SIZE %d %d %d TYPE %c %c %c (%s)\n", sz[0], sz[1], sz[2], ty[0], ty[1], ty[2], data); return 7; }
        if (ok) ++accepted; else ++rejected;
        ++combos;
      }
    std::printf("size/type combinations=%zu\n", combos);
  }
  std::printf("accepted=%zu rejected=%zu\n", accepted, rejected);
  return 0;
}
