// Probe: what do the two cross-lane maxima of attn2_fwd_kernel return?  (lane value = a distinct float per lane; -inf in some lanes)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__device__ __forceinline__ float xor16_max(float v) {
  const unsigned a = __builtin_bit_cast(unsigned, v);
  unsigned b = a;
  asm volatile("" : "+v"(b));   // a second register: with the same one on both sides the swap happens inside it and the partner's value is lost
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
  const unsigned a = __builtin_bit_cast(unsigned, v);
  unsigned b = a;
  asm volatile("" : "+v"(b));
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}
__global__ void kraw(const float* in, float* r0, float* r1) {
  const unsigned a = __builtin_bit_cast(unsigned, in[threadIdx.x]);
  unsigned b = a;
  asm volatile("" : "+v"(b));
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  r0[threadIdx.x] = __builtin_bit_cast(float, r[0]);
  r1[threadIdx.x] = __builtin_bit_cast(float, r[1]);
}
__global__ void k(const float* in, float* o16, float* o32, float* oall) {
  const float v = in[threadIdx.x];
  o16[threadIdx.x] = xor16_max(v);
  o32[threadIdx.x] = xor32_max(v);
  oall[threadIdx.x] = xor32_max(xor16_max(v));
}
int main() {
  float h[64], a[64], b[64], c[64], *d, *da, *db, *dc;
  for (int i = 0; i < 64; ++i) h[i] = (i % 16 == 5 && i / 16 == 0) ? -INFINITY : (float)((i * 37) % 101);
  hipMalloc(&d, 256); hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dc, 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, da, db, dc);
  hipMemcpy(a, da, 256, hipMemcpyDeviceToHost); hipMemcpy(b, db, 256, hipMemcpyDeviceToHost); hipMemcpy(c, dc, 256, hipMemcpyDeviceToHost);
  int bad16 = 0, bad32 = 0, badall = 0;
  for (int i = 0; i < 64; ++i) {
    const float e16 = fmaxf(h[i], h[i ^ 16]), e32 = fmaxf(h[i], h[i ^ 32]);
    const float eall = fmaxf(fmaxf(h[i & 15], h[(i & 15) + 16]), fmaxf(h[(i & 15) + 32], h[(i & 15) + 48]));
    bad16 += a[i] != e16; bad32 += b[i] != e32; badall += c[i] != eall;
  }
  printf("xor16 wrong lanes %d, xor32 wrong lanes %d, all-4 wrong lanes %d\n", bad16, bad32, badall);
  for (int i = 0; i < 64; i += 16) printf("lane %2d: in %6.1f  x16 %6.1f  x32 %6.1f  all %6.1f | lane %2d: in %6.1f x16 %6.1f x32 %6.1f all %6.1f\n", i + 5, h[i + 5], a[i + 5], b[i + 5], c[i + 5], i + 6, h[i + 6], a[i + 6], b[i + 6], c[i + 6]);
  hipLaunchKernelGGL(kraw, dim3(1), dim3(64), 0, 0, d, da, db);
  hipMemcpy(a, da, 256, hipMemcpyDeviceToHost); hipMemcpy(b, db, 256, hipMemcpyDeviceToHost);
  printf("permlane16_swap(a, copy of a): lane: in -> (first', second')\n");
  for (int i = 4; i < 64; i += 16) printf("  lane %2d: %6.1f -> (%6.1f, %6.1f)   lane %2d: %6.1f -> (%6.1f, %6.1f)\n", i, h[i], a[i], b[i], i + 2, h[i + 2], a[i + 2], b[i + 2]);
  return 0;
}
