// kernels.hip -- hand-written gfx950 (CDNA4) kernels for NVorbis' per-packet synthesis path.
//
//   k_residue        IResidue.Decode's vector adds (Residue0.cs:180-201, Residue1.cs:8-26, Residue2.cs:23-47)
//   k_couple_floor   inverse channel coupling (Mapping.cs:137-182) + IFloor.Apply
//                    (Floor1.cs:186-341, Floor0.cs:152-212)
//   k_imdct_window   IMdct.Reverse (Mdct.cs:65-535) + the window multiply of Mode.Decode (Mode.cs:160-166)
//   k_ola_*          StreamDecoder.OverlapBuffers (:532-541) + ClippingCopyBuffer / CopyBuffer (:391-415)
//
// Parity rule: every float expression keeps the reference's operand order and association and is
// rounded per operation; this file is compiled with -ffp-contract=off (no v_fma_f32 contraction), fp32
// denormals preserved.  Butterflies of one IMDCT stage touch disjoint elements, so running them in
// parallel in any order is bit-identical to the reference's sequential loops.
#include <hip/hip_runtime.h>

#include "kernels_common.h"

#define NVH_THREADS 256

__constant__ float c_inverse_db[256] = {
#include "floor1_db_table.inc"
};

// ================================================================================================
// IMDCT (generic block size n = 64 .. 8192), stage-synchronous over LDS
// ================================================================================================

// One radix-2 butterfly of "step 3" (Mdct.cs:324-329 and siblings).
__device__ __forceinline__ void bfly(float* u, int e0, int e2, float a0, float a1) {
  float k0 = u[e0] - u[e2];
  float k1 = u[e0 - 1] - u[e2 - 1];
  u[e0] = u[e0] + u[e2];
  u[e0 - 1] = u[e0 - 1] + u[e2 - 1];
  u[e2] = k0 * a0 - k1 * a1;
  u[e2 - 1] = k1 * a0 + k0 * a1;
}

// Mdct.cs:509-535
__device__ __forceinline__ void iter_54(float* e, int z) {
  float k00 = e[z] - e[z - 4];
  float y0 = e[z] + e[z - 4];
  float y2 = e[z - 2] + e[z - 6];
  float k22 = e[z - 2] - e[z - 6];
  e[z] = y0 + y2;
  e[z - 2] = y0 - y2;
  float k33 = e[z - 3] - e[z - 7];
  e[z - 4] = k00 + k33;
  e[z - 6] = k00 - k33;
  float k11 = e[z - 1] - e[z - 5];
  float y1 = e[z - 1] + e[z - 5];
  float y3 = e[z - 3] + e[z - 7];
  e[z - 1] = y1 + y3;
  e[z - 3] = y1 - y3;
  e[z - 5] = k11 - k22;
  e[z - 7] = k11 + k22;
}

// u: LDS, n/2 floats holding the spectrum on entry.  v: LDS scratch, n/2 floats (the reference's buf2).
// On return v holds the result of step 7; step 8 is done by the caller through `emit`.
template <typename Emit>
__device__ void imdct_lds(float* u, float* v, int n, const float* __restrict__ A, const float* __restrict__ B,
                          const float* __restrict__ C, const uint16_t* __restrict__ BR, int tid, int nthreads,
                          Emit emit) {
  const int n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
  int ld = 0;
  for (int t = n; t > 1; t >>= 1) ++ld;  // ilog(n) - 1

  // step 0 (Mdct.cs:74-97): spectrum (u) -> v
  for (int j = tid; j < n4; j += nthreads) {
    if (j < n8) {
      int d = n2 - 2 - 2 * j, e = 4 * j, AA = 2 * j;
      v[d + 1] = (u[e] * A[AA] - u[e + 2] * A[AA + 1]);
      v[d] = (u[e] * A[AA + 1] + u[e + 2] * A[AA]);
    } else {
      int jj = j - n8;
      int d = n4 - 2 - 2 * jj, e = n2 - 3 - 4 * jj, AA = n4 + 2 * jj;
      v[d + 1] = (-u[e + 2] * A[AA] - -u[e] * A[AA + 1]);
      v[d] = (-u[e + 2] * A[AA + 1] + -u[e] * A[AA]);
    }
  }
  __syncthreads();

  // step 2 (Mdct.cs:105-139): v -> u
  for (int c = tid; c < n8; c += nthreads) {
    int lo = 2 * c, hi = n4 + 2 * c, t = n2 - 4 - 4 * c;
    float d1 = v[hi + 1] - v[lo + 1];
    float d0 = v[hi] - v[lo];
    u[hi + 1] = v[hi + 1] + v[lo + 1];
    u[hi] = v[hi] + v[lo];
    u[lo + 1] = d1 * A[t] - d0 * A[t + 1];
    u[lo] = d0 * A[t] + d1 * A[t + 1];
  }
  __syncthreads();

  // step 3 generic stages (Mdct.cs:144-183).  Stage l: groups g < 2^(l+1), butterflies m < M_l,
  // e0 = n2-1 - g*(n>>(l+2)) - 2m, e2 = e0 - (n>>(l+3)), twiddle index m<<(l+3).  The reference runs
  // stages 0 and 1 unconditionally (:144-151), further ones while l < ld-6.
  const int L = (ld - 6) > 2 ? (ld - 6) : 2;
  for (int l = 0; l < L; ++l) {
    const int M = ((n >> (l + 4)) >> 2) << 2;
    const int total = M << (l + 1);
    for (int b = tid; b < total; b += nthreads) {
      int g = b / M, m = b - g * M;
      int e0 = n2 - 1 - g * (n >> (l + 2)) - 2 * m;
      int e2 = e0 - (n >> (l + 3));
      int t = m << (l + 3);
      bfly(u, e0, e2, A[t], A[t + 1]);
    }
    __syncthreads();
  }

  // fused last three stages (Mdct.cs:463-507)
  {
    const float A2 = A[n >> 3];
    for (int q = tid; q < (n >> 5); q += nthreads) {
      int z = n2 - 1 - 16 * q;
      float k00, k11;
      k00 = u[z] - u[z - 8];
      k11 = u[z - 1] - u[z - 9];
      u[z] = u[z] + u[z - 8];
      u[z - 1] = u[z - 1] + u[z - 9];
      u[z - 8] = k00;
      u[z - 9] = k11;

      k00 = u[z - 2] - u[z - 10];
      k11 = u[z - 3] - u[z - 11];
      u[z - 2] = u[z - 2] + u[z - 10];
      u[z - 3] = u[z - 3] + u[z - 11];
      u[z - 10] = (k00 + k11) * A2;
      u[z - 11] = (k11 - k00) * A2;

      k00 = u[z - 12] - u[z - 4];
      k11 = u[z - 5] - u[z - 13];
      u[z - 4] = u[z - 4] + u[z - 12];
      u[z - 5] = u[z - 5] + u[z - 13];
      u[z - 12] = k11;
      u[z - 13] = k00;

      k00 = u[z - 14] - u[z - 6];
      k11 = u[z - 7] - u[z - 15];
      u[z - 6] = u[z - 6] + u[z - 14];
      u[z - 7] = u[z - 7] + u[z - 15];
      u[z - 14] = (k00 + k11) * A2;
      u[z - 15] = (k00 - k11) * A2;

      iter_54(u, z);
      iter_54(u, z - 8);
    }
  }
  __syncthreads();

  // steps 4-6: bit reverse u -> v (Mdct.cs:189-214)
  for (int i = tid; i < (n >> 4); i += nthreads) {
    int d0 = n4 - 4 - 4 * i, d1 = n2 - 4 - 4 * i;
    int k4 = BR[2 * i];
    v[d1 + 3] = u[k4];
    v[d1 + 2] = u[k4 + 1];
    v[d0 + 3] = u[k4 + 2];
    v[d0 + 2] = u[k4 + 3];
    k4 = BR[2 * i + 1];
    v[d1 + 1] = u[k4];
    v[d1] = u[k4 + 1];
    v[d0 + 1] = u[k4 + 2];
    v[d0] = u[k4 + 3];
  }
  __syncthreads();

  // step 7 (Mdct.cs:217-258), in place on v
  for (int i = tid; i < (n >> 4); i += nthreads) {
    int c = 4 * i, d = 4 * i, e = n2 - 4 - 4 * i;
    float a02, a11, b0, b1, b2, b3;
    a02 = v[d] - v[e + 2];
    a11 = v[d + 1] + v[e + 3];
    b0 = C[c + 1] * a02 + C[c] * a11;
    b1 = C[c + 1] * a11 - C[c] * a02;
    b2 = v[d] + v[e + 2];
    b3 = v[d + 1] - v[e + 3];
    v[d] = b2 + b0;
    v[d + 1] = b3 + b1;
    v[e + 2] = b2 - b0;
    v[e + 3] = b1 - b3;

    a02 = v[d + 2] - v[e];
    a11 = v[d + 3] + v[e + 1];
    b0 = C[c + 3] * a02 + C[c + 2] * a11;
    b1 = C[c + 3] * a11 - C[c + 2] * a02;
    b2 = v[d + 2] + v[e];
    b3 = v[d + 3] - v[e + 1];
    v[d + 2] = b2 + b0;
    v[d + 3] = b3 + b1;
    v[e] = b2 - b0;
    v[e + 1] = b1 - b3;
  }
  __syncthreads();

  // step 8 + decode (Mdct.cs:261-312): v (buf2) -> n outputs
  for (int i = tid; i < (n >> 4); i += nthreads) {
    int b = n2 - 8 - 8 * i, e = b;
    int d0 = 4 * i, d1 = n2 - 4 - 4 * i, d2 = n2 + 4 * i, d3 = n - 4 - 4 * i;
    float p0, p1, p2, p3;
    p3 = v[e + 6] * B[b + 7] - v[e + 7] * B[b + 6];
    p2 = -v[e + 6] * B[b + 6] - v[e + 7] * B[b + 7];
    emit(d0, p3);
    emit(d1 + 3, -p3);
    emit(d2, p2);
    emit(d3 + 3, p2);

    p1 = v[e + 4] * B[b + 5] - v[e + 5] * B[b + 4];
    p0 = -v[e + 4] * B[b + 4] - v[e + 5] * B[b + 5];
    emit(d0 + 1, p1);
    emit(d1 + 2, -p1);
    emit(d2 + 1, p0);
    emit(d3 + 2, p0);

    p3 = v[e + 2] * B[b + 3] - v[e + 3] * B[b + 2];
    p2 = -v[e + 2] * B[b + 2] - v[e + 3] * B[b + 3];
    emit(d0 + 2, p3);
    emit(d1 + 1, -p3);
    emit(d2 + 2, p2);
    emit(d3 + 1, p2);

    p1 = v[e] * B[b + 1] - v[e + 1] * B[b];
    p0 = -v[e] * B[b] - v[e + 1] * B[b + 1];
    emit(d0 + 3, p1);
    emit(d1, -p1);
    emit(d2 + 3, p0);
    emit(d3, p0);
  }
}

// Stand-alone batched IMdct.Reverse: buf[b*stride .. +n) in place, no window (fine-grained ABI).
// One inverse square-polar coupling step over a pair of vectors (Mapping.cs:150-178), stand-alone (fine-grained ABI).
extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_inverse_couple(float* __restrict__ magnitude, float* __restrict__ angle, int cnt) {
  for (int j = blockIdx.x * NVH_THREADS + threadIdx.x; j < cnt; j += gridDim.x * NVH_THREADS) {
    const float oldM = magnitude[j], oldA = angle[j];
    float newM, newA;
    if (oldM > 0) {
      if (oldA > 0) { newM = oldM; newA = oldM - oldA; }
      else          { newA = oldM; newM = oldM + oldA; }
    } else {
      if (oldA > 0) { newM = oldM; newA = oldM + oldA; }
      else          { newA = oldM; newM = oldM - oldA; }
    }
    magnitude[j] = newM;
    angle[j] = newA;
  }
}

extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_mdct_reverse(float* __restrict__ buf, int n, long long stride, const float* __restrict__ A,
               const float* __restrict__ B, const float* __restrict__ C, const uint16_t* __restrict__ BR) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* u = lds;
  float* v = lds + (n >> 1);
  float* x = buf + (long long)blockIdx.x * stride;
  for (int i = threadIdx.x; i < (n >> 1); i += NVH_THREADS) u[i] = x[i];
  __syncthreads();
  imdct_lds(u, v, n, A, B, C, BR, threadIdx.x, NVH_THREADS, [=](int idx, float val) { x[idx] = val; });
}

// IMDCT + window of every ch-frame of a batch, in place on the work planes [frame][ch][block1].
extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_imdct_window(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int cf = blockIdx.x;
  const int f = cf / S.channels, c = cf - f * S.channels;
  const NvhFrame fr = Bt.frames[f];
  const int n = fr.n;
  if (n == 0) return;
  float* x = work + ((long long)f * S.channels + c) * S.block1;
  const float* __restrict__ w = S.windows + fr.window_off;
  const NvhChan chn = Bt.chans[fr.chan_off + c];
  const int tid = threadIdx.x;
  if (!chn.exec) {
    // Mapping.cs:192-196 then Mode.cs:160-166: front half keeps the residue, back half is cleared
    for (int i = tid; i < n; i += NVH_THREADS) {
      float val = (i < (n >> 1)) ? x[i] : 0.0f;
      x[i] = val * w[i];
    }
    return;
  }
  float* u = lds;
  float* v = lds + (n >> 1);
  for (int i = tid; i < (n >> 1); i += NVH_THREADS) u[i] = x[i];
  __syncthreads();
  const int s = fr.mdct_slot;
  imdct_lds(u, v, n, S.mdct_a[s], S.mdct_b[s], S.mdct_c[s], S.mdct_br[s], tid, NVH_THREADS,
            [=](int idx, float val) { x[idx] = val * w[idx]; });
}

// ================================================================================================
// Residue vector adds
// ================================================================================================

__device__ __forceinline__ void residue_apply(const NvhDevSetup& S, const NvhDevBatch& Bt, const NvhDevResidue& R,
                                              const NvhResOp& op, int i, float* planes, int half) {
  const NvhDevBook bk = S.books[op.book];
  const int dims = (int)bk.dim;
  const int offset = R.begin + (int)op.partition * R.partition_size;
  int j, comp, ch, x;
  if (R.type == 0) {
    // Residue0.cs:193-199: res[offset++] over dim-major order
    int steps = R.partition_size / dims;
    if (i >= steps * dims) return;
    comp = i / steps;
    j = i - comp * steps;
    ch = op.channel;
    x = offset + i;
  } else if (R.type == 1) {
    // Residue1.cs:19-22
    j = i / dims;
    comp = i - j * dims;
    ch = op.channel;
    x = offset + i;
  } else {
    // Residue2.cs:25-45: offset /= channels; chPtr restarts at 0 (quirk B-1)
    j = i / dims;
    comp = i - j * dims;
    ch = i % R.real_channels;
    x = offset / R.real_channels + i / R.real_channels;
  }
  unsigned e = Bt.entries[op.ent_off + j];
  if (e == NVH_ENTRY_SKIP) return;
  if (x >= half) return;  // lands in [n/2, block1): overwritten by the IMDCT or cleared, never observed
  float* p = planes + (long long)ch * S.block1 + x;
  *p = *p + S.vq[bk.tab_off + e * (unsigned)dims + (unsigned)comp];
}

extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_residue(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int clear) {
  const int f = blockIdx.x;
  const NvhFrame fr = Bt.frames[f];
  if (fr.n == 0) return;
  const int half = fr.n >> 1;
  const int tid = threadIdx.x;
  float* planes = work + (long long)f * S.channels * S.block1;
  // Array.Clear(buffer[i], 0, halfBlockSize) (Mapping.cs:108); nvh_residue_decode adds into what is there
  if (clear) {
    for (int c = 0; c < S.channels; ++c)
      for (int i = tid; i < half; i += NVH_THREADS) planes[(long long)c * S.block1 + i] = 0.0f;
    __syncthreads();
  }

  for (unsigned ps = fr.pass_begin; ps < fr.pass_end; ++ps) {
    const NvhResPass* pass = &Bt.passes[ps];
    const NvhDevResidue R = S.residues[pass->residue];
    const int psize = R.partition_size;
    for (int s = 0; s < NVH_MAX_STAGES; ++s) {
      const unsigned ob = pass->op_begin[s], oe = pass->op_begin[s + 1];
      if (ob == oe) continue;
      if (!R.sequential) {
        const long long total = (long long)(oe - ob) * psize;
        for (long long idx = tid; idx < total; idx += NVH_THREADS) {
          unsigned o = (unsigned)(idx / psize);
          int i = (int)(idx - (long long)o * psize);
          residue_apply(S, Bt, R, Bt.ops[ob + o], i, planes, half);
        }
        __syncthreads();
      } else {
        // partitions may alias (quirk B-1 / vector overrun): keep the reference's partition order
        for (unsigned o = ob; o < oe; ++o) {
          const NvhResOp op = Bt.ops[o];
          const int dims = (int)S.books[op.book].dim;
          const int cnt = ((psize + dims - 1) / dims) * dims;
          for (int i = tid; i < cnt; i += NVH_THREADS) residue_apply(S, Bt, R, op, i, planes, half);
          __syncthreads();
        }
      }
    }
  }
}

// ================================================================================================
// Inverse coupling + floor apply
// ================================================================================================

// Floor1.cs:299-314
__device__ __forceinline__ int render_point(int x0, int y0, int x1, int y1, int X) {
  int dy = y1 - y0;
  int adx = x1 - x0;
  int ady = dy < 0 ? -dy : dy;
  int err = ady * (X - x0);
  int off = err / adx;
  return dy < 0 ? y0 - off : y0 + off;
}

// Closed form of Floor1.RenderLineMulti (Floor1.cs:316-341) at abscissa x in [x0, x1):
// y = y0 + b*t + sy*floor(ady'*t/adx), t = x-x0, b = dy/adx (truncating), ady' = |dy| - |b|*adx.
__device__ __forceinline__ int line_y(int x0, int y0, int x1, int y1, int x) {
  int dy = y1 - y0;
  int adx = x1 - x0;
  int ady = dy < 0 ? -dy : dy;
  int sy = dy < 0 ? -1 : 1;
  int b = dy / adx;
  int ab = b < 0 ? -b : b;
  ady -= ab * adx;
  int t = x - x0;
  return y0 + b * t + sy * ((ady * t) / adx);
}

extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_couple_floor(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err) {
  __shared__ int s_x[NVH_MAX_POSTS + 2];
  __shared__ int s_y[NVH_MAX_POSTS + 2];
  __shared__ int s_nseg, s_flat;
  __shared__ int s_fy[NVH_MAX_POSTS];
  __shared__ int s_step[NVH_MAX_POSTS];
  __shared__ float s_coeff[256];
  const int f = blockIdx.x;
  const NvhFrame fr = Bt.frames[f];
  if (fr.n == 0) return;
  const int half = fr.n >> 1;
  const int tid = threadIdx.x;
  float* planes = work + (long long)f * S.channels * S.block1;
  const NvhChan* chans = Bt.chans + fr.chan_off;

  // inverse coupling, last step first (Mapping.cs:137-182)
  const NvhDevMapping mp = S.mappings[fr.mapping];
  for (int st = mp.coupling_steps - 1; st >= 0; --st) {
    const int mg = S.coupling[mp.coupling_off + 2 * st], an = S.coupling[mp.coupling_off + 2 * st + 1];
    if (chans[an].exec || chans[mg].exec) {
      float* M = planes + (long long)mg * S.block1;
      float* Aa = planes + (long long)an * S.block1;
      for (int j = tid; j < half; j += NVH_THREADS) {
        float oldM = M[j], oldA = Aa[j], newM, newA;
        if (oldM > 0) {
          if (oldA > 0) { newM = oldM; newA = oldM - oldA; }
          else          { newA = oldM; newM = oldM + oldA; }
        } else {
          if (oldA > 0) { newM = oldM; newA = oldM + oldA; }
          else          { newA = oldM; newM = oldM - oldA; }
        }
        M[j] = newM;
        Aa[j] = newA;
      }
    }
    __syncthreads();
  }

  for (int c = 0; c < S.channels; ++c) {
    const NvhChan chn = chans[c];
    if (!chn.exec) continue;  // uniform across the block
    float* res = planes + (long long)c * S.block1;
    const NvhDevFloor* fl = &S.floors[chn.floor];
    if (fl->type == 1) {
      const NvhDevFloor1* F = &fl->f1;
      if (chn.post_count == 0) {
        // Floor1.cs:218-221 (ForceEnergy with no posts): Array.Clear(residue, 0, n)
        for (int i = tid; i < half; i += NVH_THREADS) res[i] = 0.0f;
        continue;
      }
      __syncthreads();  // previous channel's readers are done with the shared post state
      // UnwrapPosts (Floor1.cs:224-297), parallel over dependency levels: post i only needs the final Y of
      // its two neighbours (both of lower index), so all posts of one level are independent.
      const int pc = chn.post_count;
      const uint16_t* posts = Bt.posts + chn.data_off;
      if (tid < pc) {
        s_fy[tid] = (tid < 2) ? (int)posts[tid] : 0;
        s_step[tid] = (tid < 2) ? 1 : 0;
      }
      __syncthreads();
      for (int lv = 1; lv < F->levels; ++lv) {
        if (tid >= 2 && tid < pc && F->level[tid] == lv) {
          int lo = F->l_neigh[tid], hi = F->h_neigh[tid];
          int predicted = render_point(F->x_list[lo], s_fy[lo], F->x_list[hi], s_fy[hi], F->x_list[tid]);
          int val = posts[tid];
          int highroom = F->range - predicted;
          int lowroom = predicted;
          int room = (highroom < lowroom) ? highroom * 2 : lowroom * 2;
          int fy;
          if (val != 0) {
            // stepFlags[lowOfs] = stepFlags[highOfs] = stepFlags[i] = true: only ever set, never cleared for a
            // lower index afterwards, so concurrent stores of 1 are order-free
            s_step[lo] = 1;
            s_step[hi] = 1;
            s_step[tid] = 1;
            if (val >= room) {
              if (highroom > lowroom) fy = val - lowroom + predicted;
              else fy = predicted - val + highroom - 1;
            } else {
              if ((val % 2) == 1) fy = predicted - ((val + 1) / 2);
              else fy = predicted + (val / 2);
            }
          } else {
            fy = predicted;
          }
          s_fy[tid] = fy;
        }
        __syncthreads();
      }
      // Apply's walk over the sorted posts (Floor1.cs:196-216) -> compacted list of line end points.
      // One lane of the first wavefront per sorted position (at most 64 posts).
      if (tid < 64) {
        int idx = (tid < pc) ? F->sort_idx[tid] : 0;
        bool active = (tid < pc) && s_step[idx] != 0;
        unsigned long long mask = __ballot(active);
        int rank = __popcll(mask & ((1ull << tid) - 1ull));
        int px = F->x_list[idx];
        if (active) {
          s_x[rank] = px;
          s_y[rank] = s_fy[idx] * F->multiplier;
        }
        // the walk stops at the first end point at or beyond n/2 (`if (lx >= n) break`); the line towards
        // it is drawn to min(hx, n) (quirk B-6)
        unsigned long long beyond = __ballot(active && rank >= 1 && px >= half);
        int nact = __popcll(mask);
        if (tid == 0) {
          int ns;
          if (beyond) {
            int first_lane = __ffsll((long long)beyond) - 1;
            ns = __popcll(mask & ((1ull << first_lane) - 1ull));  // rank of that end point
          } else {
            ns = nact;  // trailing flat run to n/2 (Floor1.cs:213-216)
          }
          s_nseg = ns;
          s_flat = beyond ? 0 : 1;
        }
      }
      __syncthreads();
      if (tid == 0 && s_flat) {
        int ns = s_nseg;
        s_x[ns] = half;
        s_y[ns] = s_y[ns - 1];
      }
      __syncthreads();
      const int ns = s_nseg;
      for (int x = tid; x < half; x += NVH_THREADS) {
        int k = 0;
        while (k + 1 < ns && s_x[k + 1] <= x) ++k;
        int x1 = s_x[k + 1] < half ? s_x[k + 1] : half;
        int y = line_y(s_x[k], s_y[k], x1, s_y[k + 1], x);
        if (y < 0 || y > 255) {
          atomicOr(err, NVH_DEVERR_FLOOR1_Y);  // inverse_dB_table[y] would throw (quirk B-7)
          y = y < 0 ? 0 : 255;
        }
        res[x] = res[x] * c_inverse_db[y];
      }
    } else {
      const NvhDevFloor0* F = &fl->f0;
      if (!(chn.amp > 0.0f)) {
        for (int i = tid; i < half; i += NVH_THREADS) res[i] = 0.0f;
        continue;
      }
      __syncthreads();
      // data.Coeff[i] = 2f * (float)Math.Cos(data.Coeff[i]) (Floor0.cs:165-168)
      for (int i = tid; i < F->order; i += NVH_THREADS) s_coeff[i] = 2.0f * (float)cos((double)Bt.coeffs[chn.data_off + i]);
      __syncthreads();
      const int slot = fr.mdct_slot;
      const int32_t* bark = S.ipool + F->bark_off[slot];
      const float* wmap = S.fpool + F->wmap_off[slot];
      for (int i = tid; i < half; i += NVH_THREADS) {
        int k = bark[i];
        if (k < 0 || k >= half) {
          atomicOr(err, NVH_DEVERR_FLOOR0_W);
          continue;
        }
        float p = .5f, q = .5f;
        float w = wmap[k];
        int j;
        for (j = 1; j < F->order; j += 2) {
          q = q * (w - s_coeff[j - 1]);
          p = p * (w - s_coeff[j]);
        }
        if (j == F->order) {
          q = q * (w - s_coeff[j - 1]);
          p = p * (p * (4.0f - w * w));
          q = q * q;
        } else {
          p = p * (p * (2.0f - w));
          q = q * (q * (2.0f + w));
        }
        q = chn.amp / (float)sqrt((double)(p + q)) - (float)F->amp_ofs;
        q = (float)exp((double)(q * 0.11512925f));
        res[i] = res[i] * q;
      }
    }
  }
}

// ================================================================================================
// Overlap-add + interleave + clip
// ================================================================================================


// Parallel form: valid when no overlap region reaches into a tail (FrameBatch::sequential_ola == false),
// i.e. every tail read here is an untouched windowed block.  One workgroup per frame.
extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_ola_emit(NvhDevSetup S, NvhDevBatch Bt, const float* __restrict__ work, const float* __restrict__ carry,
           float* __restrict__ pcm, int clip, int* __restrict__ clipped_flag) {
  const int f = blockIdx.x;
  const NvhFrame fr = Bt.frames[f];
  const int ch = S.channels;
  const int total = fr.emit_count * ch;
  if (total <= 0) return;
  const float* cur = work + (long long)f * ch * S.block1;
  const float* prev = nullptr;
  if (fr.ov_len > 0) prev = (fr.ov_frame == -2) ? carry : (fr.ov_frame >= 0 ? work + (long long)fr.ov_frame * ch * S.block1 : nullptr);
  float* out = pcm + fr.out_pos * ch;
  int clipped = 0;
  for (int o = threadIdx.x; o < total; o += NVH_THREADS) {
    int t = o / ch, c = o - t * ch;
    int idx = fr.emit_start + t;
    float v;
    if (fr.n == 0) {
      v = prev[(long long)c * S.block1 + fr.ov_src + t];  // drained carried tail, emitted as it is
    } else {
      v = cur[(long long)c * S.block1 + idx];
      int j = idx - fr.start;
      if (prev && j >= 0 && j < fr.ov_len) v = v + prev[(long long)c * S.block1 + fr.ov_src + j];  // OverlapBuffers
    }
    if (clip) v = clip_value(v, &clipped);
    pcm_store1(out + o, v);
  }
  report_clipped(clipped, clipped_flag);
}

// Sequential form: one workgroup walks the frames in order and performs the adds in place, exactly
// like the reference's ping-pong buffers (needed only for streams whose window flags disagree with
// their neighbours so that an overlap reaches a block's own tail).
extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_ola_emit_seq(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, const float* __restrict__ carry,
               float* __restrict__ pcm, int clip, int* __restrict__ clipped_flag) {
  const int ch = S.channels;
  int clipped = 0;
  for (int f = 0; f < Bt.nframes; ++f) {
    const NvhFrame fr = Bt.frames[f];
    float* cur = work + (long long)f * ch * S.block1;
    const float* prev = nullptr;
    if (fr.ov_len > 0) prev = (fr.ov_frame == -2) ? carry : (fr.ov_frame >= 0 ? work + (long long)fr.ov_frame * ch * S.block1 : nullptr);
    if (fr.n != 0 && prev) {
      for (int o = threadIdx.x; o < fr.ov_len * ch; o += NVH_THREADS) {
        int c = o / fr.ov_len, j = o - c * fr.ov_len;
        float* p = cur + (long long)c * S.block1 + fr.start + j;
        *p = *p + prev[(long long)c * S.block1 + fr.ov_src + j];
      }
    }
    __syncthreads();
    float* out = pcm + fr.out_pos * ch;
    const int total = fr.emit_count * ch;
    for (int o = threadIdx.x; o < total; o += NVH_THREADS) {
      int t = o / ch, c = o - t * ch;
      float v = (fr.n == 0) ? prev[(long long)c * S.block1 + fr.ov_src + t] : cur[(long long)c * S.block1 + fr.emit_start + t];
      if (clip) v = clip_value(v, &clipped);
      pcm_store1(out + o, v);
    }
    __syncthreads();
  }
  report_clipped(clipped, clipped_flag);
}


// ================================================================================================
// Overlap-add + interleave + clip from the COMPACT block layout written by k_imdct_compact
// ================================================================================================
//
// plane[0, n/4) = y[0, n/4) and plane[n/2, 3n/4) = y[n/2, 3n/4) of the un-windowed IMDCT output y; the rest follows
// from y[n/2-1-x] = -y[x] and y[n-1-x] = y[n/2+x] (Mdct.cs:275-303).  A channel that did not execute keeps its
// residue in plane[0, n/2) and has a zero tail (Mapping.cs:192-196).  The window multiply of Mode.cs:160-166
// happens here, on both operands of the overlap add, as separately rounded products.
__device__ __forceinline__ float compact_value(const float* __restrict__ plane, const float* __restrict__ w, int n,
                                               int exec, int idx) {
  const int n2 = n >> 1, n4 = n >> 2;
  float y;
  if (exec) {
    if (idx < n4) y = plane[idx];
    else if (idx < n2) y = -plane[n2 - 1 - idx];
    else if (idx < n2 + n4) y = plane[idx];
    else y = plane[n + n2 - 1 - idx];
  } else {
    y = idx < n2 ? plane[idx] : 0.0f;
  }
  return y * w[idx];
}

// Overlap-add of one frame, CH channels, everything in units of four samples: a lane produces four consecutive
// sample times of every channel and writes them as CH 16-byte stores (its 4*CH interleaved floats are contiguous).
template <int CH>
__device__ __forceinline__ int ola_vec(const NvhDevSetup& S, const NvhFrame& fr, const float* cur, const float* prev, bool prev_full,
                                       const float* __restrict__ w, const float* __restrict__ wp, float* out, int clip, int tid, int threads) {
  int clipped = 0;
  const int groups = fr.emit_count >> 2;
  for (int g = tid; g < groups; g += threads) {
    const int idx0 = fr.emit_start + 4 * g;
    const int j0 = idx0 - fr.start;
    const bool ov = prev && j0 >= 0 && j0 < fr.ov_len;  // whole group inside or outside (all multiples of 4)
    float flat[4 * CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      // exec_mask / ov_exec_mask mirror NvhChan::exec / ov_exec: no dependent load
      float4 v = compact_value4(cur + (long long)c * S.block1, w, fr.n, (fr.exec_mask >> c) & 1, idx0);
      if (ov) {
        const float* pp = prev + (long long)c * S.block1;
        const float4 t4 = prev_full ? *reinterpret_cast<const float4*>(pp + fr.ov_src + j0)
                                    : compact_value4(pp, wp, fr.ov_n, (fr.ov_exec_mask >> c) & 1, fr.ov_src + j0);
        v.x = v.x + t4.x; v.y = v.y + t4.y; v.z = v.z + t4.z; v.w = v.w + t4.w;
      }
      if (clip) {
        v.x = clip_value(v.x, &clipped); v.y = clip_value(v.y, &clipped);
        v.z = clip_value(v.z, &clipped); v.w = clip_value(v.w, &clipped);
      }
      flat[0 * CH + c] = v.x;
      flat[1 * CH + c] = v.y;
      flat[2 * CH + c] = v.z;
      flat[3 * CH + c] = v.w;
    }
    float4* o4 = reinterpret_cast<float4*>(out) + (long long)g * CH;
#pragma unroll
    for (int k = 0; k < CH; ++k) pcm_store4(o4 + k, flat[4 * k], flat[4 * k + 1], flat[4 * k + 2], flat[4 * k + 3]);
  }
  return clipped;
}


// The steady state of a stream -- a block whose whole first half overlaps the whole second half of a predecessor of the same
// size (long/long, short/short), every channel executing -- read with each compact value fetched ONCE: y[i] and y[n/2-1-i]
// of the block are +-A[i], y[n/2+i] and y[n-1-i] of the predecessor are both B[i] (Mdct.cs:275-303), so the pair of
// sample times (i, n/2-1-i) needs exactly A[i] and B[i].  A lane takes four consecutive i: two 16-byte loads per channel,
// two groups of four sample times out (ola_vec reads every value twice: once for i, once, reversed, for n/2-1-i).
// Same products, same additions, same order as ola_vec / the reference (Mode.cs:160-166, StreamDecoder.cs:532-541).
template <int CH>
__device__ __forceinline__ int ola_sym(const NvhDevSetup& S, const NvhFrame& fr, const float* cur, const float* prev,
                                       const float* __restrict__ w, const float* __restrict__ wp, float* out, int clip, int tid, int threads) {
  int clipped = 0;
  const int n = fr.n, n2 = n >> 1;
  const int groups = n >> 4;  // n/4 compact values per quarter, four per lane
  for (int g = tid; g < groups; g += threads) {
    const int i0 = 4 * g;
    const float4 wf = *reinterpret_cast<const float4*>(w + i0);                  // window at i0 .. i0+3
    const float4 wm = *reinterpret_cast<const float4*>(w + (n2 - 4 - i0));       // window at n/2-4-i0 .. n/2-1-i0
    const float4 pf = *reinterpret_cast<const float4*>(wp + (n2 + i0));          // predecessor's window at n/2+i0 ..
    const float4 pm = *reinterpret_cast<const float4*>(wp + (n - 4 - i0));       // ... and at n-4-i0 .. n-1-i0
    float fwd[4 * CH], mir[4 * CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(cur + (long long)c * S.block1 + i0);
      const float4 b = *reinterpret_cast<const float4*>(prev + (long long)c * S.block1 + n2 + i0);
      // sample times i0 .. i0+3
      float4 v = make_float4(a.x * wf.x, a.y * wf.y, a.z * wf.z, a.w * wf.w);
      float4 t = make_float4(b.x * pf.x, b.y * pf.y, b.z * pf.z, b.w * pf.w);
      v.x = v.x + t.x; v.y = v.y + t.y; v.z = v.z + t.z; v.w = v.w + t.w;
      // sample times n/2-4-i0 .. n/2-1-i0: the block's values are -A reversed, the predecessor's B reversed
      float4 u = make_float4(-a.w * wm.x, -a.z * wm.y, -a.y * wm.z, -a.x * wm.w);
      float4 r = make_float4(b.w * pm.x, b.z * pm.y, b.y * pm.z, b.x * pm.w);
      u.x = u.x + r.x; u.y = u.y + r.y; u.z = u.z + r.z; u.w = u.w + r.w;
      if (clip) {
        v.x = clip_value(v.x, &clipped); v.y = clip_value(v.y, &clipped);
        v.z = clip_value(v.z, &clipped); v.w = clip_value(v.w, &clipped);
        u.x = clip_value(u.x, &clipped); u.y = clip_value(u.y, &clipped);
        u.z = clip_value(u.z, &clipped); u.w = clip_value(u.w, &clipped);
      }
      fwd[0 * CH + c] = v.x; fwd[1 * CH + c] = v.y; fwd[2 * CH + c] = v.z; fwd[3 * CH + c] = v.w;
      mir[0 * CH + c] = u.x; mir[1 * CH + c] = u.y; mir[2 * CH + c] = u.z; mir[3 * CH + c] = u.w;
    }
    float4* of = reinterpret_cast<float4*>(out) + (long long)g * CH;
    float4* om = reinterpret_cast<float4*>(out) + (long long)((n >> 3) - 1 - g) * CH;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      pcm_store4(of + k, fwd[4 * k], fwd[4 * k + 1], fwd[4 * k + 2], fwd[4 * k + 3]);
      pcm_store4(om + k, mir[4 * k], mir[4 * k + 1], mir[4 * k + 2], mir[4 * k + 3]);
    }
  }
  return clipped;
}

// ola_sym for more than two channels: the same arithmetic, the work split per (group, channel) -- CH times the lanes of
// ola_sym, each with two 16-byte loads in flight instead of 2 CH, ~30 instead of ~100 registers -- and the interleave done
// through LDS: a lane's eight results go to channel-planar rows of the workgroup's two runs of sample times (the forward run
// [4 g0, 4 g0 + 4 GW) and the mirrored run [n/2 - 4 (g0 + GW), n/2 - 4 g0)), and after one barrier the runs leave as whole
// 16-byte vectors of interleaved, clipped PCM.  Workgroup blockIdx.y owns groups [GW y, GW y + GW).
template <int CH>
__device__ __forceinline__ int ola_sym_lds(const NvhDevSetup& S, const NvhFrame& fr, const float* cur, const float* prev,
                                           const float* __restrict__ w, const float* __restrict__ wp, float* out, int clip,
                                           float* s_run /* [2][CH][4 * NVH_OLA_GW] */) {
  constexpr int GW = NVH_OLA_GW, RUN = 4 * GW;
  const int n = fr.n, n2 = n >> 1;
  const int groups = n >> 4;
  const int g0 = (int)blockIdx.y * GW;
  if (g0 >= groups) return 0;
  const int gw = groups - g0 < GW ? groups - g0 : GW;  // groups of this workgroup (a short block has fewer than GW)
  float* sF = s_run;
  float* sM = s_run + CH * RUN;
  for (int task = threadIdx.x; task < gw * CH; task += blockDim.x) {
    const int c = task / gw, gl = task - c * gw;
    const int i0 = 4 * (g0 + gl);
    const float4 wf = *reinterpret_cast<const float4*>(w + i0);
    const float4 wm = *reinterpret_cast<const float4*>(w + (n2 - 4 - i0));
    const float4 pf = *reinterpret_cast<const float4*>(wp + (n2 + i0));
    const float4 pm = *reinterpret_cast<const float4*>(wp + (n - 4 - i0));
    const float4 a = *reinterpret_cast<const float4*>(cur + (long long)c * S.block1 + i0);
    const float4 b = *reinterpret_cast<const float4*>(prev + (long long)c * S.block1 + n2 + i0);
    float4 v = make_float4(a.x * wf.x, a.y * wf.y, a.z * wf.z, a.w * wf.w);
    const float4 t = make_float4(b.x * pf.x, b.y * pf.y, b.z * pf.z, b.w * pf.w);
    v.x = v.x + t.x; v.y = v.y + t.y; v.z = v.z + t.z; v.w = v.w + t.w;
    float4 u = make_float4(-a.w * wm.x, -a.z * wm.y, -a.y * wm.z, -a.x * wm.w);
    const float4 r = make_float4(b.w * pm.x, b.z * pm.y, b.y * pm.z, b.x * pm.w);
    u.x = u.x + r.x; u.y = u.y + r.y; u.z = u.z + r.z; u.w = u.w + r.w;
    *reinterpret_cast<float4*>(sF + c * RUN + 4 * gl) = v;              // sample times 4 g0 + 4 gl ..
    *reinterpret_cast<float4*>(sM + c * RUN + 4 * (gw - 1 - gl)) = u;   // sample times n/2 - 4 (g0 + gw) + 4 (gw - 1 - gl) ..
  }
  __syncthreads();
  int clipped = 0;
  const int nvec = gw * CH;  // 16-byte vectors per run: 4 gw sample times x CH channels
  float4* oF = reinterpret_cast<float4*>(out + (long long)(4 * g0) * CH);
  float4* oM = reinterpret_cast<float4*>(out + (long long)(n2 - 4 * (g0 + gw)) * CH);
  for (int j = threadIdx.x; j < 2 * nvec; j += blockDim.x) {
    const bool mir = j >= nvec;
    const int jj = mir ? j - nvec : j;
    const float* sr = mir ? sM : sF;
    float e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = 4 * jj + k;      // position in the run's interleaved floats
      const int tt = idx / CH, c = idx - tt * CH;
      float x = sr[c * RUN + tt];
      if (clip) x = clip_value(x, &clipped);
      e[k] = x;
    }
    pcm_store4((mir ? oM : oF) + jj, e[0], e[1], e[2], e[3]);
  }
  return clipped;
}

// A frame may be shared by gridDim.y workgroups (large frames: six channels at n = 4096 are 48 KB of PCM, and 128 lanes
// per frame leave the CUs with four wavefronts each): lane `OLA_TID` of `NVH_OLA_THREADS`.
#define NVH_OLA_THREADS ((int)(blockDim.x * gridDim.y))
#define NVH_OLA_TID ((int)(blockIdx.y * blockDim.x + threadIdx.x))
extern "C" __global__ void __launch_bounds__(256)
k_ola_compact(NvhDevSetup S, NvhDevBatch Bt, const float* __restrict__ work, const float* __restrict__ carry,
              float* __restrict__ pcm, int clip, int* __restrict__ clipped_flag, float* __restrict__ carry_out, int last_decoded,
              int nosym, const int* __restrict__ list, int emitted) {
  // list: the frames paired emission left to this kernel (nvh_launch.hip); emitted: k_synth wrote the PCM of every frame with
  // NVH_EMIT_DONE (such a frame is on the list only as the block that becomes the carried tail)
  const int f = list ? list[blockIdx.x] : (int)blockIdx.x;
  const NvhFrame fr = Bt.frames[f];
  const int ch = S.channels;
  if (f == last_decoded && carry_out) {
    // this block becomes the carried tail of the next batch (StreamDecoder's _prevPacketBuf), stored fully windowed
    const float* __restrict__ wl = S.windows + fr.window_off;
    for (int o = NVH_OLA_TID; o < (fr.n >> 2) * ch; o += NVH_OLA_THREADS) {
      int c = o / (fr.n >> 2), g = o - c * (fr.n >> 2);
      const float* plane = work + ((long long)f * ch + c) * S.block1;
      *reinterpret_cast<float4*>(carry_out + (long long)c * S.block1 + 4 * g) =
          compact_value4(plane, wl, fr.n, Bt.chans[fr.chan_off + c].exec, 4 * g);
    }
  }
  const int total = fr.emit_count * ch;
  if (total <= 0) return;
  if (emitted && (fr.emit_flags & NVH_EMIT_DONE)) return;
  const float* cur = work + (long long)f * ch * S.block1;
  const float* prev = nullptr;
  if (fr.ov_len > 0) prev = (fr.ov_frame == -2) ? carry : (fr.ov_frame >= 0 ? work + (long long)fr.ov_frame * ch * S.block1 : nullptr);
  const float* __restrict__ w = S.windows + fr.window_off;
  const float* __restrict__ wp = S.windows + fr.ov_window_off;
  const NvhChan* chans = Bt.chans + fr.chan_off;
  float* out = pcm + fr.out_pos * ch;
  int clipped = 0;
  // the carried block (ov_frame == -2) is always stored fully windowed (k_expand_carry); blocks of this batch are compact
  const bool prev_full = fr.ov_frame == -2;

  // fast path: everything in units of four samples (true for every frame of a well-formed stream except an
  // EOS-trimmed last one), up to 8 channels
  const bool vec = fr.n != 0 && ch <= 8 && ((fr.emit_start | fr.emit_count | fr.start | fr.ov_src | fr.ov_len) & 3) == 0 &&
                   ((fr.out_pos * ch) & 3) == 0;
  // steady state: whole first half over the whole second half of an executing predecessor of the same size
  const unsigned all_ch = ch >= 32 ? 0xFFFFFFFFu : ((1u << ch) - 1u);
  const bool sym = vec && prev && !prev_full && fr.ov_n == fr.n && fr.start == 0 && fr.emit_start == 0 && fr.emit_count == (fr.n >> 1) &&
                   fr.ov_src == (fr.n >> 1) && fr.ov_len == (fr.n >> 1) && (fr.exec_mask & all_ch) == all_ch &&
                   (fr.ov_exec_mask & all_ch) == all_ch && !nosym;
  if (sym && ch > 2 && gridDim.y * NVH_OLA_GW >= (unsigned)(fr.n >> 4)) {
    // more than two channels: per-(group, channel) lanes, interleave through LDS (the launch gives every frame gridDim.y
    // workgroups of NVH_OLA_GW groups each: nvh_launch.hip)
    __shared__ __attribute__((aligned(16))) float s_run[2 * 8 * 4 * NVH_OLA_GW];
    switch (ch) {
      case 3: clipped = ola_sym_lds<3>(S, fr, cur, prev, w, wp, out, clip, s_run); break;
      case 4: clipped = ola_sym_lds<4>(S, fr, cur, prev, w, wp, out, clip, s_run); break;
      case 5: clipped = ola_sym_lds<5>(S, fr, cur, prev, w, wp, out, clip, s_run); break;
      case 6: clipped = ola_sym_lds<6>(S, fr, cur, prev, w, wp, out, clip, s_run); break;
      case 7: clipped = ola_sym_lds<7>(S, fr, cur, prev, w, wp, out, clip, s_run); break;
      default: clipped = ola_sym_lds<8>(S, fr, cur, prev, w, wp, out, clip, s_run); break;
    }
    report_clipped(clipped, clipped_flag);
    return;
  }
  if (sym) {
    switch (ch) {
      case 1: clipped = ola_sym<1>(S, fr, cur, prev, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 2: clipped = ola_sym<2>(S, fr, cur, prev, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 3: clipped = ola_sym<3>(S, fr, cur, prev, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 4: clipped = ola_sym<4>(S, fr, cur, prev, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 5: clipped = ola_sym<5>(S, fr, cur, prev, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 6: clipped = ola_sym<6>(S, fr, cur, prev, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 7: clipped = ola_sym<7>(S, fr, cur, prev, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      default: clipped = ola_sym<8>(S, fr, cur, prev, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
    }
    report_clipped(clipped, clipped_flag);
    return;
  }
  if (vec) {
    switch (ch) {
      case 1: clipped = ola_vec<1>(S, fr, cur, prev, prev_full, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 2: clipped = ola_vec<2>(S, fr, cur, prev, prev_full, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 3: clipped = ola_vec<3>(S, fr, cur, prev, prev_full, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 4: clipped = ola_vec<4>(S, fr, cur, prev, prev_full, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 5: clipped = ola_vec<5>(S, fr, cur, prev, prev_full, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 6: clipped = ola_vec<6>(S, fr, cur, prev, prev_full, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      case 7: clipped = ola_vec<7>(S, fr, cur, prev, prev_full, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
      default: clipped = ola_vec<8>(S, fr, cur, prev, prev_full, w, wp, out, clip, NVH_OLA_TID, NVH_OLA_THREADS); break;
    }
    report_clipped(clipped, clipped_flag);
    return;
  }

  for (int o = NVH_OLA_TID; o < total; o += NVH_OLA_THREADS) {
    int t = o / ch, c = o - t * ch;
    int idx = fr.emit_start + t;
    const NvhChan cn = chans[c];
    float v;
    if (fr.n == 0) {
      // drained carried tail (StreamDecoder.cs:352-356): the previous block's windowed samples as they are
      v = prev[(long long)c * S.block1 + fr.ov_src + t];
    } else {
      v = compact_value(cur + (long long)c * S.block1, w, fr.n, cn.exec, idx);
      int j = idx - fr.start;
      if (prev && j >= 0 && j < fr.ov_len) {  // OverlapBuffers: next[start + j] += previous[prevStart + j]
        const float* pp = prev + (long long)c * S.block1;
        v = v + (prev_full ? pp[fr.ov_src + j] : compact_value(pp, wp, fr.ov_n, cn.ov_exec, fr.ov_src + j));
      }
    }
    if (clip) v = clip_value(v, &clipped);
    pcm_store1(out + o, v);
  }
  report_clipped(clipped, clipped_flag);
}

// Expands the compact planes of one frame into the fully windowed block (the carried tail format shared by all
// overlap kernels).  One workgroup, launched once per batch for its last decoded frame.
extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_expand_carry(NvhDevSetup S, NvhDevBatch Bt, const float* __restrict__ work, float* __restrict__ carry_out, int f) {
  const NvhFrame fr = Bt.frames[f];
  const int ch = S.channels;
  const float* __restrict__ w = S.windows + fr.window_off;
  for (int o = threadIdx.x; o < fr.n * ch; o += NVH_THREADS) {
    int c = o / fr.n, i = o - c * fr.n;
    const float* plane = work + ((long long)f * ch + c) * S.block1;
    carry_out[(long long)c * S.block1 + i] = compact_value(plane, w, fr.n, Bt.chans[fr.chan_off + c].exec, i);
  }
}

// ================================================================================================
// Stand-alone mirrors of the remaining per-packet float loops (fine-grained ABI, unit parity)
// ================================================================================================

// Plain float4 copy (grid-stride, 16 bytes per lane): the measured HBM ceiling bench.py reports next to the roofline.
extern "C" __global__ void __launch_bounds__(256)
k_copy_f4(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
  // four independent 16-byte loads per lane in flight before the first store
  const long long step = (long long)gridDim.x * 1024;
  long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  for (; i + 768 < n4; i += step) {
    const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
  }
  for (; i < n4; i += 256) dst[i] = src[i];  // the last, partial group of this workgroup's stride
}

// Mode.Decode's window loop (Mode.cs:160-166): buf[b*stride + i] *= window[i] for i < n.
extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_window_apply(float* __restrict__ buf, const float* __restrict__ window, int n, long long stride, int batch) {
  const long long total = (long long)batch * n;
  for (long long idx = (long long)blockIdx.x * NVH_THREADS + threadIdx.x; idx < total; idx += (long long)gridDim.x * NVH_THREADS) {
    const long long b = idx / n;
    const int i = (int)(idx - b * n);
    float* p = buf + b * stride + i;
    *p = *p * window[i];
  }
}

// StreamDecoder.OverlapBuffers (StreamDecoder.cs:532-541): next[c][next_start + j] += previous[c][prev_start + j].
extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_overlap_buffers(const float* __restrict__ previous, float* __restrict__ next, int prev_start, int len, int next_start,
                  int channels, long long plane_stride) {
  const long long total = (long long)channels * len;
  for (long long idx = (long long)blockIdx.x * NVH_THREADS + threadIdx.x; idx < total; idx += (long long)gridDim.x * NVH_THREADS) {
    const int c = (int)(idx / len);
    const int j = (int)(idx - (long long)c * len);
    float* p = next + c * plane_stride + next_start + j;
    *p = *p + previous[c * plane_stride + prev_start + j];
  }
}

// ClippingCopyBuffer / CopyBuffer (StreamDecoder.cs:391-415): planar -> interleaved, Utils.ClipValue when asked.
extern "C" __global__ void __launch_bounds__(NVH_THREADS)
k_copy_buffer(const float* __restrict__ planes, int start, int count, int channels, long long plane_stride,
              float* __restrict__ target, int clip, int* __restrict__ clipped_flag) {
  const long long total = (long long)count * channels;
  int clipped = 0;
  for (long long idx = (long long)blockIdx.x * NVH_THREADS + threadIdx.x; idx < total; idx += (long long)gridDim.x * NVH_THREADS) {
    const long long t = idx / channels;
    const int c = (int)(idx - t * channels);
    float v = planes[c * plane_stride + start + t];
    if (clip) v = clip_value(v, &clipped);
    target[idx] = v;
  }
  report_clipped(clipped, clipped_flag);
}
