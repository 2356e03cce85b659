// Sustained v_mfma_f32_32x32x2_f32 rate of the chip (no memory traffic): the practical ceiling under the 157.3 TFLOP/s
// paper peak (256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz).  Usage: mfma_peak [waves_per_simd] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same loop on v_mfma_f32_16x16x4_f32 (8 passes, 2048 flop): what a 16-row-owning workgroup can issue (tools/rowown_lab.hip)
template <int NACC>
__global__ __launch_bounds__(256) void mfma16_loop(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC>
void run16(int wgs_per_cu, int iters) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(mfma16_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma16_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 2048.0;
    printf("16x16x4 nacc=%d wgs/cu=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, iters, ms, flops / ms / 1e9);
  }
}
template <int NACC>
void run(int wgs_per_cu, int iters) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("nacc=%d wgs/cu=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, iters, ms, flops / ms / 1e9);
  }
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  run<4>(1, iters); run<4>(2, iters); run<2>(2, iters); run<1>(4, iters); run<4>(4, iters / 2);
  run16<8>(1, iters); run16<4>(1, iters); run16<8>(2, iters); run16<2>(4, iters);
  return 0;
}
