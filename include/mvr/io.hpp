// include/mvr/io.hpp -- the on-disk formats either side of the hot path, header-only:
//   * PCD v0.7 reader/writer for the reference's rich point (PointXYZRGBNormal, mvr/include/types.h:15):
//     `pcl::io::loadPCDFile(filename, *this)` (mvr/src/point_cloud.cpp:84) accepts DATA ascii / binary /
//     binary_compressed; `PCDWriter::writeBinaryCompressed` (point_cloud.cpp:117-118) is what the reference writes.
//     The published PCD format: text header (VERSION, FIELDS, SIZE, TYPE, COUNT, WIDTH, HEIGHT, VIEWPOINT, POINTS,
//     DATA); binary = packed records in FIELDS order; binary_compressed = u32 compressed size, u32 uncompressed
//     size, then an LZF stream of the records re-ordered field by field (all x, all y, ...).
//   * `points.asc` of saveRegisteredPoints (mvr/src/registrator.cpp:386-395): "%f %f %f %d %d %d\n" per point.
//   * the dataset tree `points/object_%05d/view_%02d/points.pcd` (mvr/src/file_system_model.cpp:286,300,313).
// PCL is not in this image, so this is a restatement of the published file format, pinned by round trips,
// hand-written files and an independent Python decoder in tests/ (parity with PCL itself: unpinned).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "types.hpp"

namespace mvr {
namespace io {

// pcl::PointXYZRGBNormal without the SSE padding
struct RichPoint {
  float x = 0, y = 0, z = 0;
  uint8_t r = 0, g = 0, b = 0;
  float normal_x = 0, normal_y = 0, normal_z = 0, curvature = 0;
};
typedef std::vector<RichPoint> RichCloud;

enum PcdMode { PCD_ASCII = 0, PCD_BINARY = 1, PCD_BINARY_COMPRESSED = 2 };

// ------------------------------------------------------------------ LZF (the byte format liblzf defines)
// control byte c < 32: c + 1 literal bytes follow;  otherwise a back reference: length = (c >> 5) + 2
// (if c >> 5 == 7 one more length byte is added), offset = ((c & 31) << 8 | next byte) + 1 behind the output.
inline size_t lzf_decompress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap)
{
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    const unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t run = ctrl + 1;
      if (ip + run > in_len || op + run > out_cap) return 0;
      std::memcpy(out + op, in + ip, run);
      ip += run; op += run;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) { if (ip >= in_len) return 0; len += in[ip++]; }
      len += 2;
      if (ip >= in_len) return 0;
      const size_t off = (((size_t)ctrl & 31) << 8 | in[ip++]) + 1;
      if (off > op || op + len > out_cap) return 0;
      for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - off];      // may overlap: byte by byte
    }
  }
  return op;
}

// greedy hash-chain-free compressor (one candidate per 3-byte hash); output is a valid LZF stream of at most
// in_len + in_len / 32 + 1 bytes
inline size_t lzf_compress(const uint8_t *in, size_t in_len, std::vector<uint8_t> &out)
{
  out.clear();
  out.reserve(in_len + in_len / 32 + 8);
  std::vector<int64_t> table(1 << 14, -1);
  size_t ip = 0, lit_start = 0;
  auto flush_literals = [&](size_t end) {
    size_t p = lit_start;
    while (p < end) {
      const size_t run = std::min<size_t>(32, end - p);
      out.push_back((uint8_t)(run - 1));
      out.insert(out.end(), in + p, in + p + run);
      p += run;
    }
  };
  while (ip + 2 < in_len) {
    const uint32_t h = ((uint32_t)in[ip] << 16 | (uint32_t)in[ip + 1] << 8 | in[ip + 2]) * 2654435761u >> 18;
    const int64_t cand = table[h];
    table[h] = (int64_t)ip;
    if (cand >= 0 && ip - (size_t)cand <= 8192 && in[cand] == in[ip] && in[cand + 1] == in[ip + 1] && in[cand + 2] == in[ip + 2]) {
      size_t len = 3;
      const size_t max_len = std::min<size_t>(264, in_len - ip);
      while (len < max_len && in[cand + len] == in[ip + len]) ++len;
      flush_literals(ip);
      const size_t off = ip - (size_t)cand - 1, l = len - 2;
      if (l < 7) out.push_back((uint8_t)((l << 5) | (off >> 8)));
      else { out.push_back((uint8_t)((7u << 5) | (off >> 8))); out.push_back((uint8_t)(l - 7)); }
      out.push_back((uint8_t)(off & 255));
      ip += len;
      lit_start = ip;
    } else {
      ++ip;
    }
  }
  flush_literals(in_len);
  return out.size();
}

// ------------------------------------------------------------------ PCD
struct PcdField { std::string name; int size = 4; char type = 'F'; int count = 1; size_t offset = 0; };

namespace detail {

inline double field_value(const uint8_t *p, const PcdField &f)
{
  switch (f.type) {
    case 'F': if (f.size == 4) { float v; std::memcpy(&v, p, 4); return v; } else { double v; std::memcpy(&v, p, 8); return v; }
    case 'U': if (f.size == 1) return *p; if (f.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; }
              if (f.size == 4) { uint32_t v; std::memcpy(&v, p, 4); return v; } { uint64_t v; std::memcpy(&v, p, 8); return (double)v; }
    default:  if (f.size == 1) return (int8_t)*p; if (f.size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
              if (f.size == 4) { int32_t v; std::memcpy(&v, p, 4); return v; } { int64_t v; std::memcpy(&v, p, 8); return (double)v; }
  }
}

inline void assign(RichPoint &pt, const PcdField &f, const uint8_t *p)
{
  if (f.name == "x") pt.x = (float)field_value(p, f);
  else if (f.name == "y") pt.y = (float)field_value(p, f);
  else if (f.name == "z") pt.z = (float)field_value(p, f);
  else if (f.name == "normal_x") pt.normal_x = (float)field_value(p, f);
  else if (f.name == "normal_y") pt.normal_y = (float)field_value(p, f);
  else if (f.name == "normal_z") pt.normal_z = (float)field_value(p, f);
  else if (f.name == "curvature") pt.curvature = (float)field_value(p, f);
  else if (f.name == "rgb" || f.name == "rgba") {      // 0x00RRGGBB packed into 4 bytes, whatever TYPE says
    uint32_t v; std::memcpy(&v, p, 4);
    pt.r = (uint8_t)(v >> 16); pt.g = (uint8_t)(v >> 8); pt.b = (uint8_t)v;
  }
}

}  // namespace detail

// pcl::io::loadPCDFile.  Returns false (cloud untouched) on a missing / malformed / truncated file or one without x y z.
inline bool loadPCDFile(const std::string &filename, RichCloud &cloud)
{
  std::ifstream f(filename.c_str(), std::ios::binary);
  if (!f) return false;
  std::vector<PcdField> fields;
  size_t points = 0, width = 0, height = 1;
  std::string data, line;
  bool have_points = false;
  while (std::getline(f, line)) {
    if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key; ss >> key;
    if (key == "FIELDS") { std::string n; while (ss >> n) { PcdField pf; pf.name = n; fields.push_back(pf); } }
    else if (key == "SIZE") { for (auto &pf : fields) if (!(ss >> pf.size)) return false; }
    else if (key == "TYPE") { for (auto &pf : fields) if (!(ss >> pf.type)) return false; }
    else if (key == "COUNT") { for (auto &pf : fields) if (!(ss >> pf.count)) return false; }
    else if (key == "WIDTH") ss >> width;
    else if (key == "HEIGHT") ss >> height;
    else if (key == "POINTS") { ss >> points; have_points = true; }
    else if (key == "DATA") { ss >> data; break; }
  }
  if (fields.empty() || data.empty()) return false;
  if (!have_points) points = width * height;
  size_t step = 0;
  bool hx = false, hy = false, hz = false;
  for (auto &pf : fields) {
    if (pf.size != 1 && pf.size != 2 && pf.size != 4 && pf.size != 8) return false;
    if (pf.count < 1 || pf.count > 4096 || (pf.type != 'F' && pf.type != 'U' && pf.type != 'I')) return false;
    // SIZE and TYPE must agree, or a decoder that trusts TYPE reads past the bytes SIZE reserved: a float is 4 or 8
    // bytes (field_value reads exactly that many), the packed colour is one 4-byte value (assign reads 4)
    if (pf.type == 'F' && pf.size != 4 && pf.size != 8) return false;
    if ((pf.name == "rgb" || pf.name == "rgba") && pf.size != 4) return false;
    pf.offset = step; step += (size_t)pf.size * pf.count;
    hx |= pf.name == "x"; hy |= pf.name == "y"; hz |= pf.name == "z";
  }
  if (!hx || !hy || !hz) return false;
  // A header is input, not truth: before any buffer is sized from it, the point count must fit the bytes that
  // actually follow (every encoding spends at least one byte per point; binary exactly `step` per point; LZF
  // expands at most 88-fold: a 3-byte reference yields up to 264 bytes).
  const std::streampos body = f.tellg();
  f.seekg(0, std::ios::end);
  const size_t remain = (body >= 0 && f.tellg() >= body) ? (size_t)(f.tellg() - body) : 0;
  f.seekg(body);
  if (step == 0 || step > (1u << 20) || points > remain) return false;
  if (data == "binary" && points > remain / step) return false;
  if (data == "binary_compressed" && points > (remain * 100 + 64) / step) return false;
  RichCloud out(points);
  if (data == "ascii") {
    for (size_t i = 0; i < points; ++i) {
      if (!std::getline(f, line)) return false;
      std::istringstream ss(line);
      for (const auto &pf : fields)
        for (int c = 0; c < pf.count; ++c) {
          uint8_t buf[8] = {0};
          if (pf.name == "rgb" || pf.name == "rgba") {
            // PCL prints a float-typed rgb as the float whose BITS are the colour; integer types print the integer
            if (pf.type == 'F') { float v; if (!(ss >> v)) return false; std::memcpy(buf, &v, 4); }
            else { uint32_t v; if (!(ss >> v)) return false; std::memcpy(buf, &v, 4); }
            PcdField raw = pf; raw.type = 'U'; raw.size = 4;
            if (c == 0) detail::assign(out[i], raw, buf);
          } else {
            double v; if (!(ss >> v)) return false;
            float fv = (float)v; std::memcpy(buf, &fv, 4);
            PcdField raw = pf; raw.type = 'F'; raw.size = 4;
            if (c == 0) detail::assign(out[i], raw, buf);
          }
        }
    }
  } else if (data == "binary") {
    std::vector<uint8_t> buf(points * step);
    if (points && !f.read(reinterpret_cast<char *>(buf.data()), (std::streamsize)buf.size())) return false;
    for (size_t i = 0; i < points; ++i)
      for (const auto &pf : fields) detail::assign(out[i], pf, &buf[i * step + pf.offset]);
  } else if (data == "binary_compressed") {
    uint32_t csize = 0, usize = 0;
    if (!f.read(reinterpret_cast<char *>(&csize), 4) || !f.read(reinterpret_cast<char *>(&usize), 4)) return false;
    if ((size_t)usize != points * step || (size_t)csize + 8 > remain || (size_t)usize > (size_t)csize * 100 + 64) return false;
    std::vector<uint8_t> cbuf(csize), ubuf(usize);
    if (csize && !f.read(reinterpret_cast<char *>(cbuf.data()), csize)) return false;
    if (usize && lzf_decompress(cbuf.data(), csize, ubuf.data(), usize) != usize) return false;
    size_t base = 0;                                     // field-major: all values of field 0, then field 1, ...
    for (const auto &pf : fields) {
      const size_t fs = (size_t)pf.size * pf.count;
      for (size_t i = 0; i < points; ++i) detail::assign(out[i], pf, &ubuf[base + i * fs]);
      base += fs * points;
    }
  } else {
    return false;
  }
  cloud.swap(out);
  return true;
}

// PCDWriter::write{ASCII,Binary,BinaryCompressed}<PointXYZRGBNormal>: FIELDS x y z rgb normal_x normal_y normal_z curvature
inline bool savePCDFile(const std::string &filename, const RichCloud &cloud, PcdMode mode = PCD_BINARY_COMPRESSED)
{
  std::ofstream f(filename.c_str(), std::ios::binary);
  if (!f) return false;
  const size_t n = cloud.size();
  f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb normal_x normal_y normal_z curvature\n"
       "SIZE 4 4 4 4 4 4 4 4\nTYPE F F F F F F F F\nCOUNT 1 1 1 1 1 1 1 1\nWIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n
    << "\nDATA " << (mode == PCD_ASCII ? "ascii" : mode == PCD_BINARY ? "binary" : "binary_compressed") << "\n";
  auto rgb_bits = [](const RichPoint &p) { return (uint32_t)p.r << 16 | (uint32_t)p.g << 8 | (uint32_t)p.b; };
  if (mode == PCD_ASCII) {
    char line[256];
    for (const RichPoint &p : cloud) {
      const uint32_t bits = rgb_bits(p); float rgbf; std::memcpy(&rgbf, &bits, 4);
      std::snprintf(line, sizeof line, "%.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", p.x, p.y, p.z, rgbf, p.normal_x, p.normal_y, p.normal_z, p.curvature);
      f << line;
    }
    return (bool)f;
  }
  const size_t step = 32;
  std::vector<uint8_t> rec(n * step);
  const bool soa = mode == PCD_BINARY_COMPRESSED;
  for (size_t i = 0; i < n; ++i) {
    const RichPoint &p = cloud[i];
    const uint32_t bits = rgb_bits(p);
    const float v[8] = {p.x, p.y, p.z, 0.f, p.normal_x, p.normal_y, p.normal_z, p.curvature};
    for (int k = 0; k < 8; ++k) {
      uint8_t *dst = soa ? &rec[((size_t)k * n + i) * 4] : &rec[i * step + (size_t)k * 4];
      if (k == 3) std::memcpy(dst, &bits, 4); else std::memcpy(dst, &v[k], 4);
    }
  }
  if (mode == PCD_BINARY) { f.write(reinterpret_cast<const char *>(rec.data()), (std::streamsize)rec.size()); return (bool)f; }
  std::vector<uint8_t> comp;
  lzf_compress(rec.data(), rec.size(), comp);
  const uint32_t csize = (uint32_t)comp.size(), usize = (uint32_t)rec.size();
  f.write(reinterpret_cast<const char *>(&csize), 4);
  f.write(reinterpret_cast<const char *>(&usize), 4);
  f.write(reinterpret_cast<const char *>(comp.data()), (std::streamsize)comp.size());
  return (bool)f;
}

// registrator.cpp:386-395: "%f %f %f %d %d %d\n"
inline bool savePointsASC(const std::string &filename, const RichCloud &cloud)
{
  FILE *file = std::fopen(filename.c_str(), "w");
  if (file == NULL) return false;
  for (const RichPoint &p : cloud) std::fprintf(file, "%f %f %f %d %d %d\n", p.x, p.y, p.z, (int)p.r, (int)p.g, (int)p.b);
  std::fclose(file);
  return true;
}

// pcl::PLYWriter::write<PointXYZ>(filename, cloud) as PointCloud::save uses it (point_cloud.cpp:99-113): ASCII PLY,
// one vertex element with float x y z, followed by the one-row `camera` element PCL's writer appends (identity axes,
// viewport = cloud width x height).  PCL is not in this image: the header is a restatement of the PLY format and of
// PCL's published field list (parity with PCL's own file: unpinned); any header-driven PLY reader accepts it.
inline bool savePLYFile(const std::string &filename, const PointCloud<PointXYZ> &cloud)
{
  FILE *file = std::fopen(filename.c_str(), "w");
  if (file == NULL) return false;
  std::fprintf(file, "ply\nformat ascii 1.0\ncomment PCL generated\nelement vertex %zu\n", cloud.size());
  std::fprintf(file, "property float x\nproperty float y\nproperty float z\n");
  std::fprintf(file, "element camera 1\nproperty float view_px\nproperty float view_py\nproperty float view_pz\n"
                     "property float x_axisx\nproperty float x_axisy\nproperty float x_axisz\n"
                     "property float y_axisx\nproperty float y_axisy\nproperty float y_axisz\n"
                     "property float z_axisx\nproperty float z_axisy\nproperty float z_axisz\n"
                     "property float focal\nproperty float scalex\nproperty float scaley\nproperty float centerx\nproperty float centery\n"
                     "property int viewportx\nproperty int viewporty\nproperty float k1\nproperty float k2\nend_header\n");
  for (size_t i = 0; i < cloud.size(); ++i) std::fprintf(file, "%.9g %.9g %.9g\n", cloud.points[i].x, cloud.points[i].y, cloud.points[i].z);
  std::fprintf(file, "0 0 0 1 0 0 0 1 0 0 0 1 0 0 0 0 0 %zu 1 0 0\n", cloud.size());
  const bool ok = std::ferror(file) == 0;
  std::fclose(file);
  return ok;
}

// the matching reader (ASCII PLY, a vertex element that has float/double x y z among its properties; other elements
// and properties are skipped).  false: malformed / binary / no x y z; the cloud is untouched.
inline bool loadPLYFile(const std::string &filename, PointCloud<PointXYZ> &cloud)
{
  std::ifstream f(filename.c_str());
  if (!f) return false;
  std::string line;
  if (!std::getline(f, line) || line.substr(0, 3) != "ply") return false;
  struct Elem { std::string name; size_t count = 0; std::vector<std::string> props; };
  std::vector<Elem> elems;
  bool ascii = false, ended = false;
  while (std::getline(f, line)) {
    if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    std::istringstream ss(line);
    std::string key; ss >> key;
    if (key == "format") { std::string fmt; ss >> fmt; ascii = fmt == "ascii"; }
    else if (key == "element") { Elem e; if (!(ss >> e.name >> e.count)) return false; elems.push_back(e); }
    else if (key == "property") { std::string t, n; if (!(ss >> t >> n) || elems.empty()) return false; if (t == "list") return false; elems.back().props.push_back(n); }
    else if (key == "end_header") { ended = true; break; }
  }
  if (!ended || !ascii) return false;
  PointCloud<PointXYZ> out;
  bool have = false;
  for (const Elem &e : elems) {
    int ix = -1, iy = -1, iz = -1;
    for (size_t k = 0; k < e.props.size(); ++k) { if (e.props[k] == "x") ix = (int)k; if (e.props[k] == "y") iy = (int)k; if (e.props[k] == "z") iz = (int)k; }
    const bool vertex = e.name == "vertex";
    if (vertex && (ix < 0 || iy < 0 || iz < 0)) return false;
    if (vertex && e.count > (1ull << 32)) return false;
    for (size_t i = 0; i < e.count; ++i) {
      if (!std::getline(f, line)) return false;
      if (!vertex) continue;
      std::istringstream ss(line);
      std::vector<double> v(e.props.size());
      for (size_t k = 0; k < v.size(); ++k) if (!(ss >> v[k])) return false;
      out.push_back(PointXYZ((float)v[(size_t)ix], (float)v[(size_t)iy], (float)v[(size_t)iz]));
    }
    have |= vertex;
  }
  if (!have) return false;
  cloud = out;
  return true;
}

// file_system_model.cpp:286,300,313
inline std::string pointsFolder(const std::string &root, int object, int view)
{
  char buf[64];
  std::snprintf(buf, sizeof buf, "/points/object_%05d/view_%02d", object, view);
  return root + buf;
}
inline std::string pointsFilename(const std::string &root, int object, int view) { return pointsFolder(root, object, view) + "/points.pcd"; }

// point_cloud.cpp:297-299: the registration path only takes x y z of the rich cloud
inline void toXYZ(const RichCloud &in, PointCloud<PointXYZ> &out)
{
  out.points.resize(in.size());
  for (size_t i = 0; i < in.size(); ++i) { out.points[i].x = in[i].x; out.points[i].y = in[i].y; out.points[i].z = in[i].z; out.points[i].data[3] = 1.0f; }
}

}  // namespace io
}  // namespace mvr
