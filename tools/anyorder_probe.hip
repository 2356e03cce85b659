// Does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let two independent kernels of one stream overlap on gfx950?
// (hip_ext.h notes the flag "is not supported on AMD GFX9xx boards" for the module launch API.)  64-workgroup spin kernels of ~20 us:
// pairs launched plain / with the flag on the second kernel; wall time per pair says whether they ran concurrently.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long long ticks, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main() {
  int* d; hipMalloc(&d, 64);
  hipStream_t st; hipStreamCreate(&st);
  const long long ticks = 2000;   // 20 us at 100 MHz
  for (int flag = 0; flag < 2; ++flag) {
    for (int rep = 0; rep < 3; ++rep) {
      hipStreamSynchronize(st);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 200; ++i) {
        hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, nullptr, nullptr, 0, ticks, d);
        hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, ticks, d + 1);
      }
      hipStreamSynchronize(st);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
      printf("flag=%d rep=%d: %.1f us per pair of 20-us kernels\n", flag, rep, us);
    }
  }
  return 0;
}
