// valu_rate.hip -- how many cycles one wave64 VALU instruction occupies a SIMD on this chip (plain f32 add / mul, packed
// f32, 32-bit integer add, v_mul_hi_u32, v_mul_lo_u32).  Planning aid for the instruction-bound spectrum kernel:
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/valu_rate tools/ubench/valu_rate.hip && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s) {
  float a[8];
  unsigned u[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { a[j] = (float)(threadIdx.x + j); u[j] = threadIdx.x * 2654435761u + j; }
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[4];
#pragma unroll
  for (int j = 0; j < 4; j++) p[j] = f2{a[2 * j], a[2 * j + 1]};
  const f2 s2{s, s};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (OP == 0) a[j] = a[j] + s;
        if (OP == 1) a[j] = a[j] * s;
        if (OP == 2) u[j] = u[j] + (unsigned)i;
        if (OP == 3) u[j] = __umulhi(u[j], 0x9E3779B9u) + 1u;
        if (OP == 4) u[j] = u[j] * 0x9E3779B1u + 1u;
        if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
      }
      if (OP == 5) {
#pragma unroll
        for (int j = 0; j < 4; j++) p[j] = p[j] + s2;
      }
    }
  }
  float acc = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) acc += a[j] + (float)u[j];
#pragma unroll
  for (int j = 0; j < 4; j++) acc += p[j].x + p[j].y;
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
static void run(const char* name, int ops_per_iter, float* d) {
  const int iters = 4096, blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr_per_simd = (double)iters * ops_per_iter * 8.0;  // 8 waves per SIMD
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("%-28s %8.3f ms  %.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, ms, cycles / wave_instr_per_simd);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0>("v_add_f32", 64, d);
  run<1>("v_mul_f32", 64, d);
  run<2>("v_add_u32", 64, d);
  run<3>("v_mul_hi_u32 (+add)", 128, d);
  run<4>("v_mul_lo_u32 (+add)", 128, d);
  run<5>("v_pk_add_f32 (2 floats)", 32, d);
  run<6>("v_mad_u32_u24", 64, d);
  hipFree(d);
  return 0;
}
