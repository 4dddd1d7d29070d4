// tools/exp_pk.hip -- micro-benchmark behind DESIGN.md's VALU notes: per-wave and per-SIMD issue rate of the
// distance inner loop written with scalar fp32 ops vs packed v_pk_{add,mul}_f32 (pairs of queries).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exp_pk.hip -o build/exp_pk && build/exp_pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int NT = 1024;

template <bool PK, int NQ>
__global__ void __launch_bounds__(256) k(const float4 *t, const float *q, float *out, int reps)
{
  __shared__ float4 T[NT];
  for (int i = threadIdx.x; i < NT; i += 256) T[i] = t[i];
  __syncthreads();
  float acc = 0.f;
  if (PK) {
    v2f qx[NQ / 2], qy[NQ / 2], qz[NQ / 2], m[NQ / 2];
    for (int j = 0; j < NQ / 2; ++j) {
      const float *p = q + (blockIdx.x * 256 + threadIdx.x) * 3 + j * 7;
      qx[j] = v2f{p[0], p[1] + 1.f}; qy[j] = v2f{p[1], p[2]}; qz[j] = v2f{p[2], p[0]}; m[j] = v2f{1e30f, 1e30f};
    }
    for (int r = 0; r < reps; ++r) {
#pragma unroll 4
      for (int i = 0; i < NT; i += 2) {
        const float4 a = T[i], b = T[i + 1];
        const v2f ax = {a.x, a.x}, ay = {a.y, a.y}, az = {a.z, a.z}, bx = {b.x, b.x}, by = {b.y, b.y}, bz = {b.z, b.z};
#pragma unroll
        for (int j = 0; j < NQ / 2; ++j) {
          v2f dx = ax - qx[j], dy = ay - qy[j], dz = az - qz[j];
          v2f da = dx * dx; da = da + dy * dy; da = da + dz * dz;
          dx = bx - qx[j]; dy = by - qy[j]; dz = bz - qz[j];
          v2f db = dx * dx; db = db + dy * dy; db = db + dz * dz;
          m[j].x = __builtin_fminf(__builtin_fminf(m[j].x, da.x), db.x);
          m[j].y = __builtin_fminf(__builtin_fminf(m[j].y, da.y), db.y);
        }
      }
      for (int j = 0; j < NQ / 2; ++j) { qx[j] += v2f{1e-3f, 1e-3f}; }
    }
    for (int j = 0; j < NQ / 2; ++j) acc += m[j].x + m[j].y;
  } else {
    float qx[NQ], qy[NQ], qz[NQ], m[NQ];
    for (int j = 0; j < NQ; ++j) {
      const float *p = q + (blockIdx.x * 256 + threadIdx.x) * 3 + j * 7;
      qx[j] = p[0]; qy[j] = p[1]; qz[j] = p[2]; m[j] = 1e30f;
    }
    for (int r = 0; r < reps; ++r) {
#pragma unroll 4
      for (int i = 0; i < NT; i += 2) {
        const float4 a = T[i], b = T[i + 1];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          float dx = a.x - qx[j], dy = a.y - qy[j], dz = a.z - qz[j];
          float da = dx * dx; da = da + dy * dy; da = da + dz * dz;
          dx = b.x - qx[j]; dy = b.y - qy[j]; dz = b.z - qz[j];
          float db = dx * dx; db = db + dy * dy; db = db + dz * dz;
          m[j] = __builtin_fminf(__builtin_fminf(m[j], da), db);
        }
      }
      for (int j = 0; j < NQ; ++j) qx[j] += 1e-3f;
    }
    for (int j = 0; j < NQ; ++j) acc += m[j];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <bool PK, int NQ>
void run(const char *name, int blocks_per_cu, const float4 *t, const float *q, float *out)
{
  const int reps = 64, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PK, NQ>), dim3(blocks), dim3(256), 0, 0, t, q, out, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((k<PK, NQ>), dim3(blocks), dim3(256), 0, 0, t, q, out, reps); hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double evals = (double)blocks * 256 * NQ * NT * reps;
  std::printf("%-8s NQ=%d waves/SIMD=%d  %.3f ms  %.3e evals/s\n", name, NQ, blocks_per_cu, ms, evals / (ms * 1e-3));
}

int main()
{
  float4 *t; float *q, *out;
  hipMalloc(&t, NT * sizeof(float4)); hipMalloc(&q, 256 * 8 * 256 * 3 * sizeof(float) + 4096); hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  std::vector<float> h(256 * 8 * 256 * 3 + 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.01f;
  hipMemcpy(q, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  hipMemcpy(t, h.data(), NT * sizeof(float4), hipMemcpyHostToDevice);
  for (int b : {1, 2, 3, 4, 6, 8}) {
    if (b == 1) { run<false, 4>("scalar", 1, t, q, out); run<true, 4>("packed", 1, t, q, out); run<false, 8>("scalar", 1, t, q, out); run<true, 8>("packed", 1, t, q, out); }
    if (b == 2) { run<false, 4>("scalar", 2, t, q, out); run<true, 4>("packed", 2, t, q, out); run<false, 8>("scalar", 2, t, q, out); run<true, 8>("packed", 2, t, q, out); }
    if (b == 3) { run<false, 4>("scalar", 3, t, q, out); run<true, 4>("packed", 3, t, q, out); run<false, 8>("scalar", 3, t, q, out); run<true, 8>("packed", 3, t, q, out); }
    if (b == 4) { run<false, 4>("scalar", 4, t, q, out); run<true, 4>("packed", 4, t, q, out); run<false, 8>("scalar", 4, t, q, out); run<true, 8>("packed", 4, t, q, out); }
    if (b == 6) { run<false, 4>("scalar", 6, t, q, out); run<true, 4>("packed", 6, t, q, out); }
    if (b == 8) { run<false, 4>("scalar", 8, t, q, out); run<true, 4>("packed", 8, t, q, out); }
  }
  return 0;
}
