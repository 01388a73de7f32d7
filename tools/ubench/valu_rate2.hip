// valu_rate2.hip -- SIMD occupancy of single VALU instructions on gfx950, every one spelled as inline asm so that the optimiser
// cannot fold the chain (valu_rate.hip's C expressions partly were).  8 wavefronts per SIMD, 8 independent chains per lane.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/valu_rate2 tools/ubench/valu_rate2.hip && gpurun_out/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPS(X) \
  X(0, "v_add_f32", "v_add_f32 %0, %0, %1") \
  X(1, "v_mul_f32", "v_mul_f32 %0, %0, %1") \
  X(2, "v_fma_f32", "v_fma_f32 %0, %0, %1, %1") \
  X(3, "v_add_u32", "v_add_u32 %0, %0, %1") \
  X(4, "v_and_b32", "v_and_b32 %0, %0, %1") \
  X(5, "v_lshrrev_b32", "v_lshrrev_b32 %0, 1, %0") \
  X(6, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1") \
  X(7, "v_add3_u32", "v_add3_u32 %0, %0, %1, %1") \
  X(8, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc") \
  X(9, "v_med3_i32", "v_med3_i32 %0, %0, 0, %1") \
  X(10, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1") \
  X(11, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %1") \
  X(12, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1") \
  X(13, "v_mul_hi_u32_u24", "v_mul_hi_u32_u24 %0, %0, %1") \
  X(14, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %1") \
  X(15, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %1") \
  X(16, "v_cmp_lt_u32 (vcc)", "v_cmp_lt_u32 vcc, %0, %1") \
  X(17, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0") \
  X(18, "v_pk_add_f32 (as 1)", "v_pk_add_f32 %2, %2, %2") \
  X(19, "v_mad_u64_u32", "v_mad_u64_u32 %2, vcc, %0, %1, %2") \
  X(20, "v_lshl_add_u64", "v_lshl_add_u64 %2, %2, 0, %2") \
  X(21, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
  X(22, "v_mov_b32", "v_mov_b32 %0, %1") \
  X(23, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1") \
  X(24, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 5") \
  X(25, "v_cndmask_b32 e64 sgpr", "v_cndmask_b32_e64 %0, %0, %1, s[20:21]") \
  X(26, "v_max_u32", "v_max_u32 %0, %0, %1") \
  X(27, "v_max3_u32", "v_max3_u32 %0, %0, %1, %1") \
  X(28, "v_ashrrev_i32", "v_ashrrev_i32 %0, 31, %0") \
  X(29, "v_add_u32 sdwa", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") \
  X(30, "v_floor_f32", "v_floor_f32 %0, %0") \
  X(31, "v_cmp_lt_u32 e64 sgpr", "v_cmp_lt_u32_e64 s[20:21], %0, %1") \
  X(32, "v_add_co + v_addc", "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %1, vcc, %1, %1, vcc") \
  X(33, "v_cmp + v_cndmask", "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc") \
  X(34, "v_xor_b32", "v_xor_b32 %0, %0, %1") \
  X(35, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %0") \
  X(36, "v_add_f32 e64 (neg)", "v_add_f32_e64 %0, %0, -%1") \
  X(37, "v_mul_f32 x4 indep", "v_mul_f32 %0, %0, %1")

template <int OP>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters, unsigned s) {
  unsigned u[8];
  unsigned long long w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { u[j] = threadIdx.x * 2654435761u + j; w[j] = u[j]; }
  unsigned t = s + threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
#define X(id, name, text) if (OP == id) asm volatile(text : "+v"(u[j]) : "v"(t), "v"(w[j]) : "vcc");
        // the 64-bit forms write operand 2
#undef X
#define X(id, name, text) if (OP == id) { if ((id >= 18 && id <= 20) || id == 32) asm volatile(text : "+v"(u[j]), "+v"(t), "+v"(w[j]) : : "vcc", "s20", "s21"); else asm volatile(text : "+v"(u[j]) : "v"(t), "v"(w[j]) : "vcc", "s20", "s21"); }
        OPS(X)
#undef X
      }
    }
  }
  unsigned acc = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) acc += u[j] + (unsigned)w[j] + (unsigned)(w[j] >> 32);
  out[blockIdx.x * 256 + threadIdx.x] = acc + t;
}

template <int OP>
static void run(const char* name, unsigned* d) {
  const int iters = 2048, blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 16, 3u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr_per_simd = (double)iters * 64 * 8.0;  // 8 waves per SIMD
  printf("%-24s %8.3f ms  %.2f ns/1000 instr  (%.2f cycles per wave64 instruction per SIMD at 2.4 GHz)\n", name, ms,
         ms * 1e6 / wave_instr_per_simd * 1000, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
}

int main() {
  unsigned* d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(unsigned));
#define X(id, name, text) run<id>(name, d);
  OPS(X)
#undef X
  hipFree(d);
  return 0;
}
