// tools/exp_ta.hip -- micro-benchmark behind DESIGN.md's notes on the grid search: what a gather costs in the texture
// addresser / L1 of a gfx950 CU, by bytes per lane, by how many lanes are active and by how the addresses fall.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_ta.hip -o build/exp_ta && build/exp_ta
// Every wave issues `reps` x 8 independent loads (addresses from a per-lane LCG, nothing waits on a loaded value until
// the end), 32 waves per CU resident: the figure printed is CU cycles per wave-level load instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct __attribute__((packed, aligned(4))) B12 { float x, y, z; };
template <int BYTES> struct Elem;
template <> struct Elem<4> { typedef float T; static __device__ float val(float v) { return v; } };
template <> struct Elem<8> { typedef float2 T; static __device__ float val(float2 v) { return v.x + v.y; } };
template <> struct Elem<12> { typedef B12 T; static __device__ float val(B12 v) { return v.x + v.y + v.z; } };
template <> struct Elem<16> { typedef float4 T; static __device__ float val(float4 v) { return v.x + v.y + v.z + v.w; } };

// pattern: 0 random lines, 1 all lanes the same address, 2 consecutive elements, 3 groups of 8 lanes share an address
template <int BYTES>
__global__ void __launch_bounds__(256) k(const char *table, uint32_t mask_elems, int pattern, unsigned long long lanes, int reps, float *out)
{
  typedef typename Elem<BYTES>::T T;
  const int lane = threadIdx.x & 63;
  const bool active = (lanes >> lane) & 1ull;
  uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.f;
  if (active)
    for (int r = 0; r < reps; ++r) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s = s * 1664525u + 1013904223u;
        uint32_t w = s;
        if (pattern == 1) w = __builtin_amdgcn_readfirstlane(s);
        uint32_t e = (w >> 8) & mask_elems;
        if (pattern == 2) e = ((__builtin_amdgcn_readfirstlane(s) >> 8) + lane) & mask_elems;
        if (pattern == 3) e = (((uint32_t)__shfl((int)s, lane & ~7, 64)) >> 8) & mask_elems;
        v[u] = *reinterpret_cast<const T *>(table + (size_t)e * 16);      // elements 16 bytes apart whatever their size
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += Elem<BYTES>::val(v[u]);
    }
  if (acc == 123.456f) out[0] = acc;
}

template <int BYTES>
static void run(const char *d_table, uint32_t elems, int pattern, unsigned long long lanes, const char *what, float *d_out, int n_cu, double mhz)
{
  const int reps = 200, blocks = n_cu * 8;      // 8 blocks x 4 waves = 32 waves per CU
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<BYTES><<<blocks, 256>>>(d_table, elems - 1, pattern, lanes, 10, d_out);
  hipEventRecord(a);
  k<BYTES><<<blocks, 256>>>(d_table, elems - 1, pattern, lanes, reps, d_out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  const double instr_per_cu = 32.0 * reps * 8;
  std::printf("%2d B/lane  %-34s table %6u KB  %7.3f ms  %6.1f cycles per load instruction per CU\n", BYTES, what, elems * 16 / 1024, ms,
              ms * 1e-3 * mhz * 1e6 / instr_per_cu);
}

int main()
{
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const double mhz = p.clockRate / 1000.0;
  std::printf("%s, %d CUs, %.0f MHz\n", p.name, p.multiProcessorCount, mhz);
  const uint32_t max_elems = 1u << 22;      // 64 MB
  std::vector<float> h((size_t)max_elems * 4, 1.0f);
  char *d; float *o;
  hipMalloc(&d, (size_t)max_elems * 16); hipMalloc(&o, 16);
  hipMemcpy(d, h.data(), (size_t)max_elems * 16, hipMemcpyHostToDevice);
  const unsigned long long all = ~0ull, half_even = 0x5555555555555555ull, quarter = 0x1111111111111111ull, first16 = 0xFFFFull, first32 = 0xFFFFFFFFull, four = 0x0001000100010001ull;
  const int n = p.multiProcessorCount;
  for (uint32_t elems : {1024u, 1u << 16, 1u << 20}) {       // 16 KB (L1), 1 MB (L2), 16 MB (L2 of all XCDs / MALL)
    run<4>(d, elems, 0, all, "random, 64 lanes", o, n, mhz);
    run<8>(d, elems, 0, all, "random, 64 lanes", o, n, mhz);
    run<12>(d, elems, 0, all, "random, 64 lanes", o, n, mhz);
    run<16>(d, elems, 0, all, "random, 64 lanes", o, n, mhz);
    run<16>(d, elems, 0, half_even, "random, 32 lanes (every 2nd)", o, n, mhz);
    run<16>(d, elems, 0, first32, "random, 32 lanes (first)", o, n, mhz);
    run<16>(d, elems, 0, quarter, "random, 16 lanes (every 4th)", o, n, mhz);
    run<16>(d, elems, 0, first16, "random, 16 lanes (first)", o, n, mhz);
    run<16>(d, elems, 0, four, "random, 4 lanes", o, n, mhz);
    run<16>(d, elems, 1, all, "one address, 64 lanes", o, n, mhz);
    run<16>(d, elems, 3, all, "8 lanes share an address", o, n, mhz);
    run<16>(d, elems, 2, all, "consecutive, 64 lanes", o, n, mhz);
    run<4>(d, elems, 2, all, "consecutive, 64 lanes", o, n, mhz);
    run<4>(d, elems, 0, quarter, "random, 16 lanes (every 4th)", o, n, mhz);
  }
  return 0;
}
