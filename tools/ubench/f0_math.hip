// f0_math.hip -- which of Floor0's double-precision library calls differ between the device (ocml) and the host (glibc)
// after rounding to float?  Floor0.cs:167 (cos), :198 (sqrt), :201 (exp).   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off f0_math.hip -o f0_math
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const float* x, float* c, float* e, float* s, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  c[i] = 2.0f * (float)cos((double)x[i]);
  e[i] = (float)exp((double)(x[i] * 0.11512925f));
  s[i] = (float)sqrt((double)(x[i] * x[i] + 1.0f));
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n), c(n), e(n), s(n);
  srand(7);
  for (int i = 0; i < n; i++) x[i] = (float)rand() / RAND_MAX * 3.1415927f * (i & 1 ? 1.f : 30.f) - (i & 2 ? 20.f : 0.f);
  float *dx, *dc, *de, *ds;
  hipMalloc(&dx, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&de, n * 4); hipMalloc(&ds, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dc, de, ds, n);
  hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(e.data(), de, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
  long mc = 0, me = 0, ms = 0;
  for (int i = 0; i < n; i++) {
    float hc = 2.0f * (float)std::cos((double)x[i]);
    float he = (float)std::exp((double)(x[i] * 0.11512925f));
    float hs = (float)std::sqrt((double)(x[i] * x[i] + 1.0f));
    mc += hc != c[i]; me += he != e[i]; ms += hs != s[i];
  }
  printf("float results that differ device vs host of %d: cos %ld  exp %ld  sqrt %ld\n", n, mc, me, ms);
  return 0;
}
