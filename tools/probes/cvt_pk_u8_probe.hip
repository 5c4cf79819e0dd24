// Probe (round 6): rounding / saturation of v_cvt_pk_u8_f32 and of v_rcp_f32 on gfx950 -- the byte plane of the split residual
// stream packs its remainders with it.   hipcc --offload-arch=gfx950 tools/probes/cvt_pk_u8_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, float* rc, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
    rc[i] = in[i] * __builtin_amdgcn_rcpf(in[i]);
  }
}
int main() {
  float h[] = {0.f, 0.4f, 0.5f, 0.6f, 1.4f, 1.5f, 1.6f, 2.5f, 3.5f, 127.5f, 128.5f, 254.4f, 254.5f, 254.6f, 255.4f, 255.5f, 300.f, 1e9f, -0.4f, -0.6f, -5.f, NAN, INFINITY, -INFINITY};
  const int n = sizeof(h) / sizeof(h[0]);
  float *d, *r; unsigned* o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4); hipMalloc(&r, n * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, r, n);
  unsigned ho[64]; float hr[64];
  hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hr, r, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g) = %u    x*rcp(x) = %g\n", h[i], ho[i], hr[i]);
  return 0;
}
