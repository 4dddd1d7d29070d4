/*
 * csrc/synth.c -- deterministic synthetic turntable scans (host only).
 *
 * The reference ships no data; BASELINE.json's configs
 * are "synthetic turntable scans".  This generator is the workload definition
 * of SURVEY.md 8(d): units mm, sensor at the origin looking +z, turntable
 * axis/pivot from mvr/src/point_cloud.cpp:102-103, a smooth star-shaped
 * object of radius ~80 mm about the pivot, view v = object rotated by
 * +v*2pi/V about (pivot, axis), N samples uniform by area on the
 * sensor-facing side, N(0, sigma^2) noise along the normal, stored as f32.
 *
 * Object (unit direction u=(a,b,c) in the object frame, centred on the pivot):
 *   r(u) = 80 + 12 (a^3 - 3 a b^2) + 6 (2 c^3 - c) + 5 a c + 4 b
 * a polynomial on the sphere: smooth everywhere, no rotational symmetry, so
 * all 6 degrees of freedom are observable.  Surface F(p) = |p| - r(p/|p|) = 0,
 * outward normal = grad F / |grad F| (analytic).
 */
#include "mvr_hip.h"

#include <math.h>
#include <string.h>

#define API __attribute__((visibility("default")))

typedef struct { uint64_t s[4]; } xo_t;

static uint64_t splitmix64(uint64_t *x)
{
  uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t xo_next(xo_t *g)
{ /* xoshiro256** */
  uint64_t *s = g->s, r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
  return r;
}
static double xo_u01(xo_t *g) { return (double)(xo_next(g) >> 11) * (1.0 / 9007199254740992.0); }

static double radius(const double u[3], double g[3])
{
  double a = u[0], b = u[1], c = u[2];
  if (g) {
    g[0] = 12.0 * (3 * a * a - 3 * b * b) + 5.0 * c;
    g[1] = 12.0 * (-6 * a * b) + 4.0;
    g[2] = 6.0 * (6 * c * c - 1) + 5.0 * a;
  }
  return 80.0 + 12.0 * (a * a * a - 3 * a * b * b) + 6.0 * (2 * c * c * c - c) + 5.0 * a * c + 4.0 * b;
}

static void rodrigues(const double axis[3], double ang, double R[9])
{
  double n = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  double x = axis[0] / n, y = axis[1] / n, z = axis[2] / n, c = cos(ang), s = sin(ang), k = 1 - c;
  R[0] = c + x * x * k;     R[1] = x * y * k - z * s; R[2] = x * z * k + y * s;
  R[3] = y * x * k + z * s; R[4] = c + y * y * k;     R[5] = y * z * k - x * s;
  R[6] = z * x * k - y * s; R[7] = z * y * k + x * s; R[8] = c + z * z * k;
}

API void mvr_synth_default(mvr_synth_params *p, int n_views, int config_id)
{
  p->n_views = n_views;
  p->seed = 0x4D5652ull + 1000ull * (uint64_t)config_id;
  p->noise_sigma = 0.1;
  /* mvr/src/point_cloud.cpp:102-103 */
  p->pivot[0] = -13.382786; p->pivot[1] = 50.223461; p->pivot[2] = 917.4776;
  p->axis[0] = -0.054323; p->axis[1] = -0.814921; p->axis[2] = -0.577020;
}

API void mvr_synth_prior(const mvr_synth_params *p, double pivot[3], double axis[3])
{
  pivot[0] = p->pivot[0] + 1.5; pivot[1] = p->pivot[1] - 1.0; pivot[2] = p->pivot[2] + 2.0;
  double n = sqrt(p->axis[0] * p->axis[0] + p->axis[1] * p->axis[1] + p->axis[2] * p->axis[2]);
  double a[3] = { p->axis[0] / n, p->axis[1] / n, p->axis[2] / n };
  double t = 0.5 * M_PI / 180.0, c = cos(t), s = sin(t);
  axis[0] = a[0]; axis[1] = c * a[1] - s * a[2]; axis[2] = s * a[1] + c * a[2];
}

API int mvr_synth_view(const mvr_synth_params *p, int view, size_t n, float *xyzw, float *normals)
{
  if (!p || !xyzw || p->n_views < 1 || view < 0 || view >= p->n_views) return MVR_E_ARG;
  xo_t g; uint64_t sm = p->seed + (uint64_t)view;
  for (int k = 0; k < 4; ++k) g.s[k] = splitmix64(&sm);
  double R[9];
  rodrigues(p->axis, (double)view * (2.0 * M_PI / (double)p->n_views), R);
  const double wmax = 104.5 * 104.5 * 2.0;   /* bound on dA/dOmega = r^2 |grad F| */
  size_t got = 0;
  while (got < n) {
    double c = 2.0 * xo_u01(&g) - 1.0, ph = 2.0 * M_PI * xo_u01(&g), acc = xo_u01(&g);
    double sn = sqrt(1.0 - c * c), u[3] = { sn * cos(ph), sn * sin(ph), c }, gp[3];
    double r = radius(u, gp);
    double gu = gp[0] * u[0] + gp[1] * u[1] + gp[2] * u[2];
    double gf[3] = { u[0] - (gp[0] - gu * u[0]) / r, u[1] - (gp[1] - gu * u[1]) / r, u[2] - (gp[2] - gu * u[2]) / r };
    double gn = sqrt(gf[0] * gf[0] + gf[1] * gf[1] + gf[2] * gf[2]);
    if (acc * wmax > r * r * gn) continue;                 /* uniform by area */
    double no[3] = { gf[0] / gn, gf[1] / gn, gf[2] / gn }, po[3] = { r * u[0], r * u[1], r * u[2] };
    double nw[3], pw[3];
    for (int k = 0; k < 3; ++k) {
      nw[k] = R[3 * k] * no[0] + R[3 * k + 1] * no[1] + R[3 * k + 2] * no[2];
      pw[k] = R[3 * k] * po[0] + R[3 * k + 1] * po[1] + R[3 * k + 2] * po[2] + p->pivot[k];
    }
    if (nw[0] * pw[0] + nw[1] * pw[1] + nw[2] * pw[2] >= 0.0) continue;  /* faces away from the sensor */
    /* Box-Muller */
    double u1 = xo_u01(&g), u2 = xo_u01(&g);
    if (u1 < 1e-300) u1 = 1e-300;
    double e = p->noise_sigma * sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
    xyzw[4 * got] = (float)(pw[0] + e * nw[0]); xyzw[4 * got + 1] = (float)(pw[1] + e * nw[1]);
    xyzw[4 * got + 2] = (float)(pw[2] + e * nw[2]); xyzw[4 * got + 3] = 1.0f;
    if (normals) { normals[4 * got] = (float)nw[0]; normals[4 * got + 1] = (float)nw[1]; normals[4 * got + 2] = (float)nw[2]; normals[4 * got + 3] = 0.0f; }
    ++got;
  }
  return MVR_OK;
}
