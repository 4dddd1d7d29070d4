// tools/exp_gate.hip -- what a host <-> device hand-off costs on this box, behind DESIGN.md section 5 (the pipelined
// ring step): a chain of small kernels per pass whose inputs (poses) come from a host solve of the previous pass.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_gate.hip -o build/exp_gate && build/exp_gate
// Variants of ONE loop (pass = `chain` kernels of `kus` microseconds each, then a host "solve" of `solve_us`):
//   sync      : launch chain, hipStreamSynchronize, solve                       (the round-2 step)
//   spin      : launch chain, last kernel writes a flag in mapped host memory, host spins on it, solve
//   gate-wv   : pass k+1's chain is enqueued while pass k runs, behind hipStreamWaitValue32 on signal memory / on mapped
//               host memory; host spins on the completion flag, solves, releases the gate with a plain store
//   gate-spin : the same with a one-thread kernel that spins on a mapped host word (bounded: it gives up after 2 s)
// Printed: microseconds per pass minus (chain * kus + solve_us) = what the hand-off costs.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void busy_kernel(unsigned us, volatile uint32_t *flag, uint32_t seq, const double *params, double *sink)
{
  const unsigned long long t0 = wall_clock64();                  // 100 MHz
  double acc = params ? params[threadIdx.x & 15] : 0.0;           // (reads the "poses": device-visible host memory)
  while (wall_clock64() - t0 < (unsigned long long)us * 100ull) acc += 1e-9;
  if (sink && acc == 12345.0) *sink = acc;
  if (flag && blockIdx.x == 0 && threadIdx.x == 0) { __threadfence_system(); *flag = seq; }
}

__global__ void gate_kernel(volatile uint32_t *gate, uint32_t want, uint32_t *gave_up)
{
  const unsigned long long t0 = wall_clock64();
  while (*gate < want) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > 200000000ull) { if (gave_up) *gave_up = 1; break; }      // 2 s: never hang the queue
  }
}

static double now_us()
{
  using namespace std::chrono;
  return duration<double, std::micro>(steady_clock::now().time_since_epoch()).count();
}
static void host_busy(double us) { const double t0 = now_us(); while (now_us() - t0 < us) {} }

int main(int argc, char **argv)
{
  const int chain = argc > 1 ? std::atoi(argv[1]) : 10, kus = argc > 2 ? std::atoi(argv[2]) : 30, solve_us = argc > 3 ? std::atoi(argv[3]) : 75;
  const int passes = argc > 4 ? std::atoi(argv[4]) : 200;
  CK(hipSetDevice(0));
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint32_t *h_flag = nullptr, *d_flag = nullptr, *h_gate = nullptr, *d_gate = nullptr, *sig = nullptr, *gave = nullptr;
  double *h_par = nullptr, *d_par = nullptr, *sink = nullptr;
  CK(hipHostMalloc(reinterpret_cast<void **>(&h_flag), 64, hipHostMallocMapped));
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_flag), h_flag, 0));
  CK(hipHostMalloc(reinterpret_cast<void **>(&h_gate), 64, hipHostMallocMapped));
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_gate), h_gate, 0));
  CK(hipHostMalloc(reinterpret_cast<void **>(&h_par), 4096, hipHostMallocMapped));
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_par), h_par, 0));
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&gave, 64)); CK(hipMemset(gave, 0, 64));
  const hipError_t es = hipExtMallocWithFlags(reinterpret_cast<void **>(&sig), 8, hipMallocSignalMemory);
  std::printf("hipExtMallocWithFlags(hipMallocSignalMemory) -> %s, ptr %p\n", hipGetErrorString(es), (void *)sig);
  std::memset(h_par, 0, 4096);
  const double ideal = (double)chain * kus + solve_us;
  auto launch_chain = [&](uint32_t seq, bool flag_in_kernel) {
    for (int k = 0; k < chain; ++k)
      hipLaunchKernelGGL(busy_kernel, dim3(256), dim3(64), 0, s, (unsigned)kus, (k == chain - 1 && flag_in_kernel) ? d_flag : nullptr, seq, d_par, sink);
  };
  // warm-up
  for (int k = 0; k < 20; ++k) { launch_chain(0, false); CK(hipStreamSynchronize(s)); }

  {   // ---- sync
    const double t0 = now_us();
    for (int p = 0; p < passes; ++p) { launch_chain(0, false); CK(hipStreamSynchronize(s)); host_busy(solve_us); h_par[0] = p; }
    const double per = (now_us() - t0) / passes;
    std::printf("sync      : %.1f us per pass, hand-off cost %.1f us (ideal %.0f)\n", per, per - ideal, ideal);
  }
  {   // ---- spin on a flag written by the last kernel
    *h_flag = 0;
    const double t0 = now_us();
    for (int p = 0; p < passes; ++p) {
      launch_chain((uint32_t)p + 1, true);
      while (*(volatile uint32_t *)h_flag != (uint32_t)p + 1) {}
      host_busy(solve_us); h_par[0] = p;
    }
    const double per = (now_us() - t0) / passes;
    CK(hipStreamSynchronize(s));
    std::printf("spin      : %.1f us per pass, hand-off cost %.1f us\n", per, per - ideal);
  }
  {   // ---- spin on a flag written by hipStreamWriteValue32 after the chain
    *h_flag = 0;
    bool ok = true;
    const double t0 = now_us();
    for (int p = 0; p < passes && ok; ++p) {
      launch_chain(0, false);
      if (hipStreamWriteValue32(s, d_flag, (uint32_t)p + 1, 0) != hipSuccess) { ok = false; break; }
      while (*(volatile uint32_t *)h_flag != (uint32_t)p + 1) {}
      host_busy(solve_us); h_par[0] = p;
    }
    const double per = (now_us() - t0) / passes;
    CK(hipStreamSynchronize(s));
    if (ok) std::printf("spin-wrval: %.1f us per pass, hand-off cost %.1f us\n", per, per - ideal);
    else { (void)hipGetLastError(); std::printf("spin-wrval: hipStreamWriteValue32 on mapped host memory not supported\n"); }
  }
  // ---- gated: pass p+1 enqueued behind a gate while pass p runs
  for (int variant = 0; variant < 3; ++variant) {
    const char *name = variant == 0 ? "gate-wv-signal" : variant == 1 ? "gate-wv-host" : "gate-spin";
    if (variant == 0 && (es != hipSuccess || !sig)) { std::printf("%s: no signal memory\n", name); continue; }
    volatile uint32_t *hgate = variant == 0 ? sig : h_gate;      // (signal memory is host-accessible)
    uint32_t *dgate = variant == 0 ? sig : d_gate;
    *hgate = 0; *h_flag = 0;
    bool ok = true;
    auto enqueue_gated = [&](uint32_t p) {      // pass p (1-based) waits for gate >= p
      if (variant < 2) { if (hipStreamWaitValue32(s, dgate, p, hipStreamWaitValueGte, 0xFFFFFFFFu) != hipSuccess) ok = false; }
      else hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, s, (volatile uint32_t *)dgate, p, gave);
      launch_chain(p, true);
    };
    *hgate = 1;
    enqueue_gated(1);
    const double t0 = now_us();
    for (int p = 1; p <= passes && ok; ++p) {
      if (p < passes) enqueue_gated((uint32_t)p + 1);                 // while pass p runs
      while (*(volatile uint32_t *)h_flag != (uint32_t)p) {}          // pass p done
      host_busy(solve_us); h_par[0] = p;                              // the solve writes the next poses ...
      __atomic_store_n((uint32_t *)hgate, (uint32_t)p + 1, __ATOMIC_RELEASE);      // ... and releases pass p+1
    }
    const double per = (now_us() - t0) / passes;
    if (!ok) { (void)hipGetLastError(); *hgate = 0xFFFFFFFFu; std::printf("%s: hipStreamWaitValue32 failed\n", name); }
    CK(hipStreamSynchronize(s));
    uint32_t g = 0; CK(hipMemcpy(&g, gave, 4, hipMemcpyDeviceToHost));
    if (ok) std::printf("%-14s: %.1f us per pass, hand-off cost %.1f us%s\n", name, per, per - ideal, g ? "  (a gate gave up!)" : "");
  }
  return 0;
}
