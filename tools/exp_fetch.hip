// tools/exp_fetch.hip -- what rocprofv3's FETCH_SIZE counts on gfx950 for the access patterns of the grid search, against a
// known byte count (the guide's note: FETCH_SIZE reports exactly half the bytes of a wide coalesced stream; "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 tools/exp_fetch.hip -o build/exp_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- build/exp_fetch
// Three kernels over a 2 GiB table (far beyond the 256 MiB Infinity Cache), each touching every byte it reads ONCE:
//   stream16   lane i reads the 16 bytes at 16 i                      (coalesced: 1 KiB per wave, eight 128-byte lines)
//   gather16   lane i reads 16 bytes at the start of line perm(i)      (one distinct 128-byte line per lane)
//   gather64   lane i reads the 64 bytes at the start of line perm(i)  (four 16-byte loads in one line)
// The program prints the bytes each kernel asked for and the distinct lines it touched; FETCH_SIZE (KiB) per kernel comes
// from the counter file: counted bytes per touched line = what a gather's miss costs in this counter.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// a bijection on [0, 2^bits): odd multiplier + xor-shift (every line is touched exactly once)
__device__ __forceinline__ uint32_t perm(uint32_t i, int bits)
{
  const uint32_t mask = (1u << bits) - 1u;
  uint32_t x = (i * 2654435761u) & mask;
  x ^= x >> (bits / 2);
  x = (x * 40503u + 12345u * 2u + 1u) & mask;      // (odd multiplier, any addend: still a bijection modulo 2^bits)
  return x;
}

__global__ void stream16(const float4 *__restrict__ t, size_t n, float *out)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = t[i];
  if (v.x == 12345.f) out[0] = v.y;
}
__global__ void gather16(const char *__restrict__ t, uint32_t lines, int bits, float *out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lines) return;
  const float4 v = *reinterpret_cast<const float4 *>(t + (size_t)perm(i, bits) * 128);
  if (v.x == 12345.f) out[0] = v.y;
}
__global__ void gather64(const char *__restrict__ t, uint32_t lines, uint32_t first, int bits, float *out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lines) return;
  const float4 *p = reinterpret_cast<const float4 *>(t + (size_t)perm(first + i, bits) * 128);
  const float4 a = p[0], b = p[1], c = p[2], d = p[3];
  if (a.x + b.x + c.x + d.x == 12345.f) out[0] = a.y;
}

int main()
{
  const int bits = 24;                         // 2^24 lines of 128 bytes = 2 GiB
  const uint32_t lines = 1u << bits;
  const size_t bytes = (size_t)lines * 128;
  char *t = nullptr; float *out = nullptr;
  CK(hipMalloc(&t, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(t, 0, bytes));
  CK(hipDeviceSynchronize());
  const size_t n16 = bytes / 16 / 8;           // stream an eighth of the table (256 MiB asked for)
  hipLaunchKernelGGL(stream16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<const float4 *>(t) + (bytes / 16 / 2), n16, out);
  CK(hipDeviceSynchronize());
  const uint32_t g = lines / 4;                // a quarter of the lines each (disjoint halves would do; the permutation spreads them anyway)
  hipLaunchKernelGGL(gather16, dim3((g + 255) / 256), dim3(256), 0, 0, t, g, bits, out);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(gather64, dim3((g + 255) / 256), dim3(256), 0, 0, t, g, g, bits, out);      // (the NEXT quarter of the permuted lines: none of them touched before)
  CK(hipDeviceSynchronize());
  std::printf("stream16: %zu bytes asked = %zu lines\n", n16 * 16, n16 * 16 / 128);
  std::printf("gather16: %zu bytes asked, %u distinct lines (%zu line bytes)\n", (size_t)g * 16, g, (size_t)g * 128);
  std::printf("gather64: %zu bytes asked, %u distinct lines (%zu line bytes)\n", (size_t)g * 64, g, (size_t)g * 128);
  return 0;
}
