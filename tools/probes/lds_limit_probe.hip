// What is the largest dynamic LDS a kernel launch accepts on this device?  (round 6: a 163 840-byte launch was refused)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) { extern __shared__ int s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (threadIdx.x == 0) out[0] = s[1]; }
int main() {
  int v = 0; hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0); printf("MaxSharedMemoryPerBlock %d\n", v);
  int* d; hipMalloc(&d, 4);
  for (int sz : {65536, 131072, 148480, 155648, 159744, 163328, 163584, 163712, 163776, 163840}) {
    hipError_t e1 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, sz);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), sz, 0, d);
    hipError_t e2 = hipGetLastError(); hipError_t e3 = hipDeviceSynchronize();
    printf("%d: attr %s | launch %s | sync %s\n", sz, hipGetErrorName(e1), hipGetErrorName(e2), hipGetErrorName(e3));
  }
  return 0;
}
