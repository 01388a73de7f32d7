// uncached_handoff.hip -- does a kernel see what the kernel before it (same stream) stored into hipDeviceMallocUncached memory?
// The round-4 experiment that kept the work planes in uncached memory produced wrong PCM for a reason that was never found; the
// library's own consumers read stale plane data there (tools/repro_uncached.py: NaN poison written by hipMemsetAsync before the
// producer kernel comes back out of k_ola_compact), only after some allocate / free history, and never with ordinary hipMalloc
// memory.  This program has no library in it: rounds of (allocate uncached block of a size that varies like the library's work
// planes do, producer kernel stores a round-specific pattern with 16-byte stores from every CU, consumer kernel -- next launch,
// same stream -- compares), blocks freed and re-allocated between rounds, ordinary cached allocations churned next to them.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/uncached_handoff.hip -o /tmp/uncached_handoff && /tmp/uncached_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void producer(uint4* p, size_t nvec, unsigned tag) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(tag, (unsigned)i, tag ^ (unsigned)i, 0x5EED0000u + tag);
}

// frame-shaped like the library: workgroup b reads the block of workgroup b + 1 and b - 1 (written by other CUs / XCDs)
__global__ void consumer(const uint4* p, size_t nvec, unsigned tag, unsigned* bad, unsigned* stale_prev, unsigned prev_tag) {
  const size_t per = 1024;  // 16 KB "planes"
  const size_t nb = nvec / per;
  for (size_t b = blockIdx.x; b < nb; b += gridDim.x) {
    const size_t nbr = (b + 1 == nb) ? 0 : b + 1;
    for (size_t k = threadIdx.x; k < per; k += blockDim.x) {
      const size_t i = nbr * per + k;
      const uint4 v = p[i];
      if (v.x != tag || v.y != (unsigned)i || v.z != (tag ^ (unsigned)i) || v.w != 0x5EED0000u + tag) {
        atomicAdd(bad, 1u);
        if (v.x == prev_tag) atomicAdd(stale_prev, 1u);
      }
    }
  }
}

__global__ void touch(uint4* p, size_t nvec) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(0xC0FFEEu, 0xC0FFEEu, 0xC0FFEEu, 0xC0FFEEu);
}

int main(int argc, char** argv) {
  const bool uncached = !(argc > 1 && argv[1][0] == 'c');  // "c": the same with ordinary cached memory
  const bool poison = argc > 2;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned* cnt;
  CK(hipMalloc((void**)&cnt, 8));
  std::vector<void*> churn;
  const size_t sizes[] = {7 * 16384, 7 * 8192, 7 * 16384, 7 * 16384, 25 * 8192, 310 * 8192, 366 * 16384, 606 * 16384, 25 * 8192, 310 * 8192, 366 * 16384, 606 * 16384};
  unsigned total_bad = 0, prev_tag = 0;
  for (int round = 0; round < 48; ++round) {
    const size_t bytes = ((sizes[round % 12] + 256 + 4095) / 4096) * 4096;
    // cached blocks come and go next to the uncached one (the library's slabs, PCM, staging)
    void* c1; CK(hipMalloc(&c1, bytes / 2 + 4096));
    hipLaunchKernelGGL(touch, dim3(512), dim3(256), 0, st, (uint4*)c1, (bytes / 2) / 16);
    churn.push_back(c1);
    if (churn.size() > 3) { CK(hipStreamSynchronize(st)); CK(hipFree(churn.front())); churn.erase(churn.begin()); }
    void* p;
    if (uncached) CK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached)); else CK(hipMalloc(&p, bytes));
    const unsigned tag = 0x1000u + (unsigned)round;
    const size_t nvec = bytes / 16;
    CK(hipMemsetAsync(cnt, 0, 8, st));
    if (poison) CK(hipMemsetAsync(p, 0xFF, bytes, st));
    for (int rep = 0; rep < 4; ++rep) {  // a batch is synthesised more than once in the library too
      hipLaunchKernelGGL(producer, dim3(2048), dim3(256), 0, st, (uint4*)p, nvec, tag + 0x100u * rep);
      hipLaunchKernelGGL(consumer, dim3(2048), dim3(256), 0, st, (const uint4*)p, nvec, tag + 0x100u * rep, cnt, cnt + 1, rep ? tag + 0x100u * (rep - 1) : prev_tag);
    }
    unsigned h[2];
    CK(hipMemcpyAsync(h, cnt, 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    if (h[0]) printf("round %2d: %zu bytes at %p: %u of %zu 16-byte vectors stale or wrong in the consumer (%u hold the previous pattern)\n", round, bytes, p, h[0], nvec * 4, h[1]);
    total_bad += h[0];
    prev_tag = tag + 0x300u;
    CK(hipFree(p));
  }
  printf("%s memory%s: %u vectors the consumer kernel did not see as the producer kernel stored them\n", uncached ? "hipDeviceMallocUncached" : "hipMalloc", poison ? " (poisoned by hipMemsetAsync first)" : "", total_bad);
  return 0;
}
