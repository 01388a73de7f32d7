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

#ifdef NVH_EXPERIMENTS  // k_imdct_ola measured slower than k_spectrum_imdct + k_ola_compact: experiments build only
// ================================================================================================
// Fused IMDCT + window + overlap-add + interleave + clip  (block sizes 256 .. 2048, up to 4 channels)
// ================================================================================================
//
//   IMdct.Reverse + window          Mdct.cs:65-313, Mode.cs:160-166
//   OverlapBuffers                  StreamDecoder.cs:532-541
//   ClippingCopyBuffer / CopyBuffer StreamDecoder.cs:391-415, Utils.cs:30-43
//
// One workgroup = one RUN of consecutive frames, one wavefront per channel.  A wave walks its channel through
// the run keeping the previous block's windowed second half (the only part a later frame can overlap with) in
// its LDS slice, so the windowed blocks never travel through HBM: the kernel reads n/2 spectrum floats and
// writes (valid - start) PCM floats per channel-frame -- exactly the algorithmic traffic of SURVEY 8d -- plus
// one recomputed "halo" frame per run (the frame before the run's first, needed for its tail; the same trick
// as the reference's one-packet pre-roll after a seek, StreamDecoder.cs:602-623).
// Interleaving: each wave stages its channel's emitted samples planar in LDS (its IMDCT scratch is free by
// then), the workgroup then writes [t][c] with 16-byte stores.
//
// Host-checked preconditions (nvh_api.hip, otherwise the unfused kernels run): every overlap lands inside the
// first half of its block and comes from the second half of the previous one; block sizes 256..2048.

namespace {

template <int LD>
__device__ __forceinline__ void block_chunks(const NvhDevSetup& S, const NvhFrame& fr, bool exec, const float* X, float* buf,
                                             int lane, float4 (&ov)[8], int (&oi)[8]) {
  using G = Geo<LD>;
  const float* __restrict__ w = S.windows + fr.window_off;
  const int sl = fr.mdct_slot;
  if (exec) {
#pragma unroll
    for (int k = 0; k < 8; ++k) oi[k] = -1;
    imdct_wave_fast<LD, true>(X, w, buf, S.mdct_a[sl], S.mdct_b[sl], S.mdct_c[sl], S.mdct_tw[sl], lane,
                              [&](int slot, int idx, float4 v) { ov[slot] = v; oi[slot] = idx; });
  } else {
    // Mapping.cs:192-196 + Mode.cs:160-166: front half keeps the residue, back half is cleared, all windowed
    const int p = lane;
    const bool on = p < (G::n >> 5);
    const int i8[2] = {p, (G::n >> 4) - 1 - p};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int d[4] = {4 * i8[h], G::n2 - 4 - 4 * i8[h], G::n2 + 4 * i8[h], G::n - 4 - 4 * i8[h]};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        oi[4 * h + q] = on ? d[q] : -1;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on && d[q] < G::n2) x = *reinterpret_cast<const float4*>(X + d[q]);
        float4 ww = on ? *reinterpret_cast<const float4*>(w + d[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
        ov[4 * h + q] = make_float4(x.x * ww.x, x.y * ww.y, x.z * ww.z, x.w * ww.w);
      }
    }
  }
}

__device__ __forceinline__ float clip1(float v, int& clipped) {  // Utils.cs:30-43
  if (v > .99999994f) { clipped = 1; return 0.99999994f; }
  if (v < -.99999994f) { clipped = 1; return -0.99999994f; }
  return v;
}

}  // namespace

namespace {

// One frame of one channel inside a run: block synthesis, overlap-add with the tail left by the previous
// frame, new tail, staging + interleaved emission.  emit == false: halo frame (only its tail is produced).
// Kept out of line per block size so that the kernel's register allocation is that of one instance, not
// the union of all of them.
template <int LD>
__device__ __attribute__((noinline)) int ola_frame(const NvhDevSetup& S, const NvhDevBatch& Bt, int f, bool emit,
                                                   const float* __restrict__ work, float* __restrict__ carry_out,
                                                   float* __restrict__ pcm, int clip, int last_decoded, float* lds,
                                                   int stride, int BUF) {
  const int nch = S.channels;
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x, nthr = nch * 64;
  float* buf = lds + c * stride;
  float* tail = buf + BUF;
  const NvhFrame fr = Bt.frames[f];
  const bool exec = Bt.chans[fr.chan_off + c].exec != 0;
  const float* X = work + ((long long)f * nch + c) * S.block1;
  int clipped = 0;
  float4 ov[8];
  int oi[8];
  block_chunks<LD>(S, fr, exec, X, buf, lane, ov, oi);
  const int h2 = fr.n >> 1;
  if (!emit) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (oi[k] >= h2) *reinterpret_cast<float4*>(tail + (oi[k] - h2)) = ov[k];
    wave_sync();
    return 0;
  }
  float* out = pcm + fr.out_pos * nch;
  // phase 1: OverlapBuffers -- next[start + j] += previous[prevStart + j]; reads the old tail only
  if (fr.ov_len > 0) {
    const int toff = fr.ov_src - (fr.ov_n >> 1) - fr.start;  // tail index = idx + toff
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = oi[k];
      if (idx < 0 || idx >= h2) continue;
      const int j = idx - fr.start;
      if (j >= 0 && j + 3 < fr.ov_len && (toff & 3) == 0) {
        const float4 t4 = *reinterpret_cast<const float4*>(tail + idx + toff);
        ov[k].x = ov[k].x + t4.x; ov[k].y = ov[k].y + t4.y; ov[k].z = ov[k].z + t4.z; ov[k].w = ov[k].w + t4.w;
      } else if (j + 3 >= 0 && j < fr.ov_len) {
        if (j >= 0 && j < fr.ov_len) ov[k].x = ov[k].x + tail[idx + toff];
        if (j + 1 >= 0 && j + 1 < fr.ov_len) ov[k].y = ov[k].y + tail[idx + 1 + toff];
        if (j + 2 >= 0 && j + 2 < fr.ov_len) ov[k].z = ov[k].z + tail[idx + 2 + toff];
        if (j + 3 >= 0 && j + 3 < fr.ov_len) ov[k].w = ov[k].w + tail[idx + 3 + toff];
      }
    }
  }
  wave_sync();
  // phase 2: new tail, staging of the emitted range (or direct stores when it does not fit the scratch)
  const bool direct = fr.emit_count > BUF;
  const bool aligned = ((fr.emit_start | fr.emit_count) & 3) == 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = oi[k];
    if (idx < 0) continue;
    if (idx >= h2) {
      *reinterpret_cast<float4*>(tail + (idx - h2)) = ov[k];
      if (f == last_decoded) *reinterpret_cast<float4*>(carry_out + (long long)c * S.block1 + idx) = ov[k];
    }
    const int t = idx - fr.emit_start;
    if (t + 3 < 0 || t >= fr.emit_count) continue;
    const float e4[4] = {ov[k].x, ov[k].y, ov[k].z, ov[k].w};
    if (!direct && aligned && t >= 0 && t + 3 < fr.emit_count) {
      *reinterpret_cast<float4*>(buf + t) = ov[k];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int te = t + e;
        if (te < 0 || te >= fr.emit_count) continue;
        if (direct) {
          float v = e4[e];
          if (clip) v = clip1(v, clipped);
          out[(long long)te * nch + c] = v;
        } else {
          buf[te] = e4[e];
        }
      }
    }
  }
  __syncthreads();
  // interleave + clip: [t][c], 16-byte stores when the geometry allows
  if (!direct) {
    const bool vec = aligned && ((fr.out_pos * nch) & 3) == 0;
    if (vec && nch == 2) {
      const float* s0 = lds;
      const float* s1 = lds + stride;
      for (int g = tid; g < (fr.emit_count >> 2); g += nthr) {
        float4 a = reinterpret_cast<const float4*>(s0)[g], b = reinterpret_cast<const float4*>(s1)[g];
        if (clip) {
          a.x = clip1(a.x, clipped); a.y = clip1(a.y, clipped); a.z = clip1(a.z, clipped); a.w = clip1(a.w, clipped);
          b.x = clip1(b.x, clipped); b.y = clip1(b.y, clipped); b.z = clip1(b.z, clipped); b.w = clip1(b.w, clipped);
        }
        reinterpret_cast<float4*>(out)[2 * g] = make_float4(a.x, b.x, a.y, b.y);
        reinterpret_cast<float4*>(out)[2 * g + 1] = make_float4(a.z, b.z, a.w, b.w);
      }
    } else if (vec && nch == 1) {
      for (int g = tid; g < (fr.emit_count >> 2); g += nthr) {
        float4 a = reinterpret_cast<const float4*>(lds)[g];
        if (clip) { a.x = clip1(a.x, clipped); a.y = clip1(a.y, clipped); a.z = clip1(a.z, clipped); a.w = clip1(a.w, clipped); }
        reinterpret_cast<float4*>(out)[g] = a;
      }
    } else {
      for (int o = tid; o < fr.emit_count * nch; o += nthr) {
        int t = o / nch, cc = o - t * nch;
        float v = lds[cc * stride + t];
        if (clip) v = clip1(v, clipped);
        out[o] = v;
      }
    }
  }
  __syncthreads();
  return clipped;
}

__device__ __forceinline__ int ola_frame_any(int n, const NvhDevSetup& S, const NvhDevBatch& Bt, int f, bool emit,
                                             const float* work, float* carry_out, float* pcm, int clip, int last_decoded,
                                             float* lds, int stride, int BUF) {
  switch (n) {
    case 256: return ola_frame<8>(S, Bt, f, emit, work, carry_out, pcm, clip, last_decoded, lds, stride, BUF);
    case 512: return ola_frame<9>(S, Bt, f, emit, work, carry_out, pcm, clip, last_decoded, lds, stride, BUF);
    case 1024: return ola_frame<10>(S, Bt, f, emit, work, carry_out, pcm, clip, last_decoded, lds, stride, BUF);
    default: return ola_frame<11>(S, Bt, f, emit, work, carry_out, pcm, clip, last_decoded, lds, stride, BUF);
  }
}

}  // namespace

extern "C" __global__ void __launch_bounds__(256)
k_imdct_ola(NvhDevSetup S, NvhDevBatch Bt, const float* __restrict__ work, const float* __restrict__ carry_in,
            float* __restrict__ carry_out, float* __restrict__ pcm, int clip, int* __restrict__ clipped_flag, int run_len,
            int last_decoded) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nch = S.channels;
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x, nthr = nch * 64;
  const int BUF = 2 * ((S.block1 >> 2) + (S.block1 >> 5));  // Geo<>::LDS_FLOATS of the largest block
  const int stride = BUF + (S.block1 >> 1);
  float* tail = lds + c * stride + BUF;  // previous block's windowed second half
  const int f0 = blockIdx.x * run_len;
  const int f1 = (f0 + run_len) < Bt.nframes ? (f0 + run_len) : Bt.nframes;
  int clipped = 0;

  // ---- tail of the frame before the run ----
  {
    const NvhFrame fr0 = Bt.frames[f0];
    if (fr0.n != 0 && fr0.ov_len > 0) {
      if (fr0.ov_frame == -2) {
        const float* src = carry_in + (long long)c * S.block1 + (fr0.ov_n >> 1);
        for (int j = lane; j < (fr0.ov_n >> 3); j += 64) reinterpret_cast<float4*>(tail)[j] = reinterpret_cast<const float4*>(src)[j];
        wave_sync();
      } else if (fr0.ov_frame >= 0) {
        ola_frame_any(fr0.ov_n, S, Bt, fr0.ov_frame, false, work, carry_out, pcm, clip, last_decoded, lds, stride, BUF);
      }
    }
  }

  for (int f = f0; f < f1; ++f) {
    const int n = Bt.frames[f].n;
    if (n == 0) {
      // drained carried tail (StreamDecoder.cs:352-356): emitted as it is
      const NvhFrame fr = Bt.frames[f];
      float* out = pcm + fr.out_pos * nch;
      for (int o = tid; o < fr.emit_count * nch; o += nthr) {
        int t = o / nch, cc = o - t * nch;
        float v = carry_in[(long long)cc * S.block1 + fr.ov_src + t];
        if (clip) v = clip1(v, clipped);
        out[o] = v;
      }
      continue;
    }
    clipped |= ola_frame_any(n, S, Bt, f, true, work, carry_out, pcm, clip, last_decoded, lds, stride, BUF);
  }
  report_clipped(clipped, clipped_flag);
}
#endif  // NVH_EXPERIMENTS
