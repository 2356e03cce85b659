// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds element (row, col) = row * 256 + col of a [64][128] bf16-sized (u16) image with plain 256-byte rows;
// every lane supplies the address of (row q, columns 4 p ..) of its 16-lane group's 4 x 16 block at (row0 = 0, col0 = 16 * group) and prints what it got.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out) {
  __shared__ unsigned short img[64 * 128];
  for (int i = threadIdx.x; i < 64 * 128; i += 64) img[i] = (unsigned short)((i / 128) * 256 + (i % 128));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const unsigned addr = (unsigned)(uintptr_t)img + (unsigned)((q * 128 + 16 * g + 4 * pp) * 2);
  u16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf("  (r%d,c%3d)", h[l * 4 + e] / 256, h[l * 4 + e] % 256);
    printf("\n");
  }
  return 0;
}
