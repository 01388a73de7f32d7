// kernels_imdct.hip -- wavefront-per-channel-frame IMDCT + window for gfx950 (block sizes 256..8192).
// The transform itself lives in imdct_wave.h; this file holds the kernels that run it from / to global memory.
#include <hip/hip_runtime.h>

#include "imdct_wave.h"
#include "kernels_common.h"

// One 64-lane workgroup per channel-frame.  In place on the work planes [frame][ch][block1].
extern "C" __global__ void __launch_bounds__(64)
k_imdct_wave(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int cf = blockIdx.x;
  const int f = cf / S.channels, c = cf - f * S.channels;
  const NvhFrame fr = Bt.frames[f];
  const int n = fr.n;
  if (n == 0) return;
  float* x = work + ((long long)f * S.channels + c) * S.block1;
  const float* __restrict__ w = S.windows + fr.window_off;
  const NvhChan chn = Bt.chans[fr.chan_off + c];
  const int lane = threadIdx.x;
  if (!chn.exec) {
    // Mapping.cs:192-196 then Mode.cs:160-166
    for (int i = lane; i < n; i += 64) {
      float val = (i < (n >> 1)) ? x[i] : 0.0f;
      x[i] = val * w[i];
    }
    return;
  }
  const int s = fr.mdct_slot;
  const float* A = S.mdct_a[s];
  const float* B = S.mdct_b[s];
  const float* C = S.mdct_c[s];
  const float* TW = S.mdct_tw[s];
  switch (n) {
    case 256: imdct_wave<8, true>(x, x, w, lds, A, B, C, TW, lane); break;
    case 512: imdct_wave<9, true>(x, x, w, lds, A, B, C, TW, lane); break;
    case 1024: imdct_wave<10, true>(x, x, w, lds, A, B, C, TW, lane); break;
    case 2048: imdct_wave<11, true>(x, x, w, lds, A, B, C, TW, lane); break;
    case 4096: imdct_wave<12, true>(x, x, w, lds, A, B, C, TW, lane); break;
    case 8192: imdct_wave<13, true>(x, x, w, lds, A, B, C, TW, lane); break;
    default: break;  // 64 / 128: handled by the generic kernel (host never launches this one for them)
  }
}

// Compact variant: plane[0, n/4) and plane[n/2, 3n/4) receive the un-windowed independent quarters
// (see imdct_wave<.., COMPACT>); consumed by k_ola_compact.
extern "C" __global__ void __launch_bounds__(64)
k_imdct_compact(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int cf = blockIdx.x;
  const int f = cf / S.channels, c = cf - f * S.channels;
  const NvhFrame fr = Bt.frames[f];
  const int n = fr.n;
  if (n == 0) return;
  float* x = work + ((long long)f * S.channels + c) * S.block1;
  const NvhChan chn = Bt.chans[fr.chan_off + c];
  const int lane = threadIdx.x;
  if (!chn.exec) {
    // the residue stays in [0, n/2) (nothing to exploit; k_ola_compact windows it); its tail quarter is zero
    for (int i = lane; i < (n >> 2); i += 64) x[(n >> 1) + i] = 0.0f;
    return;
  }
  const int s = fr.mdct_slot;
  const float* A = S.mdct_a[s];
  const float* B = S.mdct_b[s];
  const float* C = S.mdct_c[s];
  const float* TW = S.mdct_tw[s];
  switch (n) {
    case 256: imdct_wave<8, false, true>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    case 512: imdct_wave<9, false, true>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    case 1024: imdct_wave<10, false, true>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    case 2048: imdct_wave<11, false, true>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    case 4096: imdct_wave<12, false, true>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    case 8192: imdct_wave<13, false, true>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    default: break;
  }
}

// Stand-alone batched IMdct.Reverse on the wave path (fine-grained ABI, n >= 256).
extern "C" __global__ void __launch_bounds__(64)
k_mdct_reverse_wave(float* __restrict__ buf, int n, long long stride, const float* __restrict__ A,
                    const float* __restrict__ B, const float* __restrict__ C, const float* __restrict__ TW) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* x = buf + (long long)blockIdx.x * stride;
  const int lane = threadIdx.x;
  switch (n) {
    case 256: imdct_wave<8, false>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    case 512: imdct_wave<9, false>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    case 1024: imdct_wave<10, false>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    case 2048: imdct_wave<11, false>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    case 4096: imdct_wave<12, false>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    case 8192: imdct_wave<13, false>(x, x, nullptr, lds, A, B, C, TW, lane); break;
    default: break;
  }
}

