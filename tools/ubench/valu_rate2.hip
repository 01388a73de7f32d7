// valu_rate2.hip -- cycles a wave64 VALU / LDS instruction occupies a SIMD on this chip, measured with s_memtime inside the kernel
// (no clock assumption) and with every measured instruction written in inline assembly (nothing for the optimiser to fold):
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/valu_rate2 tools/ubench/valu_rate2.hip && gpurun_out/valu_rate2
// 8 workgroups of 4 wavefronts per CU (8 wavefronts per SIMD, the spectrum kernel's occupancy) and 1 wavefront per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X

template <int OP>
__global__ void __launch_bounds__(256) k(long long* out, int iters, unsigned seed) {
  unsigned r0 = threadIdx.x * 2654435761u + seed, r1 = r0 ^ 0x12345u, r2 = r0 + 77u, r3 = r0 * 3u;
  unsigned c0 = seed | 0x10001u, c1 = 0x3f800001u;
  __shared__ unsigned lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 4u;
  __syncthreads();
  unsigned a0 = (threadIdx.x * 4u) & 16380u;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    // four independent chains x 8 = 32 instructions per iteration
    if (OP == 0) { REP8(asm volatile("v_add_f32 %0, %0, %5\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1));) }
    if (OP == 1) { REP8(asm volatile("v_mul_f32 %0, %0, %5\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1));) }
    if (OP == 2) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1));) }
    if (OP == 3) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1));) }
    if (OP == 4) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1));) }
    if (OP == 5) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_mad_u32_u24 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1));) }
    if (OP == 6) { REP8(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1));) }
    if (OP == 7) { REP8(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1));) }
    if (OP == 8) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1) : "vcc");) }
    if (OP == 9) { REP8(asm volatile("v_sub_f32 %0, %0, %5\n v_mul_f32 %1, %1, %5\n v_mul_hi_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1));) }
    if (OP == 10) {  // dependent chain of LDS reads: latency
      REP8(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a0) : : "memory");)
    }
    if (OP == 11) {  // independent LDS reads: throughput
      REP8(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0) : "memory");)
    }
    if (OP == 12) {  // a dependent VALU chain (one wave's view of back-to-back dependent issue)
      REP8(asm volatile("v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c0), "v"(c1));)
    }
  }
  const long long t1 = clock64();
  if (OP == 10) r0 = a0;
  out[blockIdx.x * 256 + threadIdx.x] = (t1 - t0) + ((r0 ^ r1 ^ r2 ^ r3) == 0x5a5a5a5au ? 1 : 0);
}

template <int OP>
static void run(const char* name, long long* d, std::vector<long long>& h) {
  const int iters = 2048;
  for (int waves_per_simd : {8, 1}) {
    const int blocks = 256 * waves_per_simd;  // 4 wavefronts per workgroup: one per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 16, 1u);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(long long) * (size_t)blocks * 256, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks * 256; i += 64) sum += (double)h[i];
    const double per_wave = sum / (blocks * 4);
    const double instr = (double)iters * 32.0;
    printf("%-44s %d wave/SIMD: %7.2f cycles per instruction in a wave, %6.2f per instruction and SIMD\n", name, waves_per_simd,
           per_wave / instr, per_wave / (instr * waves_per_simd));
  }
}

int main() {
  long long* d;
  const size_t n = (size_t)256 * 8 * 256;
  hipMalloc(&d, n * sizeof(long long));
  std::vector<long long> h(n);
  run<0>("v_add_f32", d, h);
  run<1>("v_mul_f32", d, h);
  run<2>("v_add_u32", d, h);
  run<3>("v_mul_hi_u32", d, h);
  run<4>("v_mul_lo_u32", d, h);
  run<5>("v_mad_u32_u24", d, h);
  run<6>("v_and_b32", d, h);
  run<7>("v_lshrrev_b32", d, h);
  run<8>("v_cndmask_b32", d, h);
  run<9>("mix: sub_f32 / mul_f32 / mul_hi_u32 / add_u32", d, h);
  run<12>("v_add_f32, dependent chain", d, h);
  run<10>("ds_read_b32, dependent chain (latency)", d, h);
  run<11>("ds_read_b32, 4 in flight", d, h);
  hipFree(d);
  return 0;
}
