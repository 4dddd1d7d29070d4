// include/mvr/types.hpp -- value types of the drop-in C++ shim.
//
// The reference compiles its registration call sites against PCL and Eigen
// types (mvr/include/types.h:14-18: PCLPoint = pcl::PointXYZ, PCLPointCloud =
// pcl::PointCloud<PCLPoint>; Eigen::Matrix4f from icp.getFinalTransformation(),
// mvr/src/registrator.cpp:573).  Neither library exists in this image, so the
// shim supplies self-contained types with the same names, members and memory
// layout; with -DMVR_ALIAS_PCL they are also reachable as pcl:: / Eigen::.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

namespace mvr {

// pcl::PointXYZ: 16 bytes, data[3] is padding that PCL keeps at 1.0f.
struct alignas(16) PointXYZ {
  union {
    float data[4];
    struct { float x, y, z; };
  };
  PointXYZ() : data{0.f, 0.f, 0.f, 1.f} {}
  PointXYZ(float _x, float _y, float _z) : data{_x, _y, _z, 1.f} {}
};
static_assert(sizeof(PointXYZ) == 16, "PointXYZ must match pcl::PointXYZ");

// pcl::PointCloud<PointT> (the members the reference touches: points, size,
// at, operator[], push_back, clear, operator+=, Ptr; registrator.cpp:548-576,
// point_cloud.cpp:290-303).
template <typename PointT>
class PointCloud {
 public:
  typedef std::shared_ptr<PointCloud<PointT> > Ptr;
  typedef std::shared_ptr<const PointCloud<PointT> > ConstPtr;
  std::vector<PointT> points;
  uint32_t width = 0, height = 1;
  bool is_dense = true;

  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = 0; height = 1; }
  void reserve(size_t n) { points.reserve(n); }
  void resize(size_t n) { points.resize(n); width = (uint32_t)n; height = 1; }
  void push_back(const PointT &p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
  PointT &at(size_t i) { return points.at(i); }
  const PointT &at(size_t i) const { return points.at(i); }
  PointT &operator[](size_t i) { return points[i]; }
  const PointT &operator[](size_t i) const { return points[i]; }
  PointCloud &operator+=(const PointCloud &rhs)
  {
    points.insert(points.end(), rhs.points.begin(), rhs.points.end());
    width = (uint32_t)points.size(); height = 1;
    is_dense = is_dense && rhs.is_dense;
    return *this;
  }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};

// pcl::Correspondence {index_query, index_match, distance}; `distance` holds
// the SQUARED distance, as PCL's correspondence estimation stores it.
struct Correspondence {
  int index_query = 0;
  int index_match = -1;
  float distance = 0.f;
  Correspondence() {}
  Correspondence(int q, int m, float d) : index_query(q), index_match(m), distance(d) {}
};
typedef std::vector<Correspondence> Correspondences;
typedef std::shared_ptr<Correspondences> CorrespondencesPtr;

// ---- small fixed matrices with Eigen's storage: column-major, operator()(row, col)
template <typename S>
struct Mat4 {
  S m[16];
  Mat4() { setIdentity(); }
  explicit Mat4(const S *colmajor) { std::memcpy(m, colmajor, sizeof m); }
  static Mat4 Identity() { return Mat4(); }
  static Mat4 Zero() { Mat4 r; for (S &v : r.m) v = S(0); return r; }
  void setIdentity() { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? S(1) : S(0); }
  S &operator()(int r, int c) { return m[r + 4 * c]; }
  S operator()(int r, int c) const { return m[r + 4 * c]; }
  S coeff(int r, int c) const { return m[r + 4 * c]; }
  S *data() { return m; }
  const S *data() const { return m; }
  // C = A * B, terms added in k order, each operation rounded in S
  Mat4 operator*(const Mat4 &b) const
  {
    Mat4 r;
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < 4; ++i) {
        S s = m[i] * b.m[4 * j];
        s = s + m[i + 4] * b.m[4 * j + 1];
        s = s + m[i + 8] * b.m[4 * j + 2];
        s = s + m[i + 12] * b.m[4 * j + 3];
        r.m[i + 4 * j] = s;
      }
    return r;
  }
  template <typename T>
  Mat4<T> cast() const { Mat4<T> r; for (int i = 0; i < 16; ++i) r.m[i] = (T)m[i]; return r; }
  bool isIdentity() const { for (int i = 0; i < 16; ++i) if (m[i] != ((i % 5 == 0) ? S(1) : S(0))) return false; return true; }
};
typedef Mat4<float> Matrix4f;
typedef Mat4<double> Matrix4d;

// Eigen::Affine3f as returned by LUM::getTransformation (registrator.cpp:658):
// only .data() / .matrix() are used by the reference.
struct Affine3f {
  Matrix4f mat;
  float *data() { return mat.data(); }
  const float *data() const { return mat.data(); }
  const Matrix4f &matrix() const { return mat; }
};

struct Vector6f {
  float v[6] = {0, 0, 0, 0, 0, 0};
  float &operator()(int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
};

// The scene-graph side of the reference stores poses in osg::Matrix: double,
// ROW-vector convention (v' = v * M).  RowMatrixd mirrors that type so the
// reference's pose algebra reads the same: `A * B` = "A then B",
// preMult(v) = v * M (point_cloud.cpp:298, registrator.cpp:574).
struct RowMatrixd {
  double m[4][4];
  RowMatrixd() { makeIdentity(); }
  static RowMatrixd identity() { return RowMatrixd(); }
  void makeIdentity() { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m[r][c] = (r == c); }
  bool isIdentity() const { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) if (m[r][c] != (r == c ? 1.0 : 0.0)) return false; return true; }
  double &operator()(int r, int c) { return m[r][c]; }
  double operator()(int r, int c) const { return m[r][c]; }
  RowMatrixd operator*(const RowMatrixd &b) const
  {
    RowMatrixd r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = m[i][0] * b.m[0][j];
        s = s + m[i][1] * b.m[1][j];
        s = s + m[i][2] * b.m[2][j];
        s = s + m[i][3] * b.m[3][j];
        r.m[i][j] = s;
      }
    return r;
  }
  // column-vector 4x4, column-major: element (r,c) = m[c][r]; as a flat array it
  // is exactly this object's row-major storage.
  const double *asColumnMajorColumnVector() const { return &m[0][0]; }
  static RowMatrixd translate(double x, double y, double z) { RowMatrixd r; r.m[3][0] = x; r.m[3][1] = y; r.m[3][2] = z; return r; }
  // osg::Matrix::rotate(from, to): the shortest rotation that takes direction `from` to direction `to` (v' = v * M)
  static RowMatrixd rotateFromTo(double fx, double fy, double fz, double tx, double ty, double tz)
  {
    const double fn = std::sqrt(fx * fx + fy * fy + fz * fz), tn = std::sqrt(tx * tx + ty * ty + tz * tz);
    fx /= fn; fy /= fn; fz /= fn; tx /= tn; ty /= tn; tz /= tn;
    const double cx = fy * tz - fz * ty, cy = fz * tx - fx * tz, cz = fx * ty - fy * tx;
    const double sn = std::sqrt(cx * cx + cy * cy + cz * cz), cs = fx * tx + fy * ty + fz * tz;
    if (sn < 1e-12) {
      if (cs > 0) return RowMatrixd();
      // opposite directions: half a turn about any axis orthogonal to `from`
      double ax = 0, ay = -fz, az = fy;
      if (std::fabs(fx) < std::fabs(fy) && std::fabs(fx) < std::fabs(fz)) { ax = 0; ay = -fz; az = fy; }
      else if (std::fabs(fy) < std::fabs(fz)) { ax = -fz; ay = 0; az = fx; }
      else { ax = -fy; ay = fx; az = 0; }
      return rotate(3.14159265358979323846, ax, ay, az);
    }
    return rotate(std::atan2(sn, cs), cx, cy, cz);
  }
  // right-handed rotation by `angle` about `axis`, acting as v' = v * M
  static RowMatrixd rotate(double angle, double ax, double ay, double az)
  {
    const double n = std::sqrt(ax * ax + ay * ay + az * az);
    const double x = ax / n, y = ay / n, z = az / n, c = std::cos(angle), s = std::sin(angle), k = 1.0 - c;
    const double R[3][3] = {{c + x * x * k, x * y * k - z * s, x * z * k + y * s},
                            {y * x * k + z * s, c + y * y * k, y * z * k - x * s},
                            {z * x * k - y * s, z * y * k + x * s, c + z * z * k}};
    RowMatrixd r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[j][i] = R[i][j];   // transpose: row-vector form
    return r;
  }
};

// mvr/include/types.h:20-50 PclMatrixCaster: the transposing bridge between a
// row-vector scene matrix and the column-vector Matrix4f.
template <class Matrix>
class PclMatrixCaster {
 public:
  PclMatrixCaster(const Matrix4f &m) : m_(m) {}
  PclMatrixCaster(const Matrix &m) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m_(i, j) = (float)m(j, i); }
  operator Matrix4f() const { return m_; }
  operator Matrix() const
  {
    Matrix m;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m(i, j) = m_(j, i);
    return m;
  }
 private:
  Matrix4f m_;
};

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};

}  // namespace mvr
