// kernels_spectrum2.hip -- gather-form spectrum kernel for the common stream shape (1 or 2 channels, one
// submap, residue type 1 or 2 without partition aliasing).
//
//   Array.Clear + IResidue.Decode adds   Mapping.cs:108,133; Residue1.cs:8-26, Residue2.cs:23-47
//   inverse square-polar coupling         Mapping.cs:137-182
//   IFloor.Apply                          Floor1.cs:186-341 (UnwrapPosts :224-297), Floor0.cs:152-212
//
// Where k_spectrum (kernels_spectrum.hip) replays the reference's scatter -- one barrier per residue stage,
// read-modify-write of an LDS spectrum, a coupling pass, a floor pass -- this kernel turns the frame inside out:
// one thread owns one frequency bin of ALL channels and
//   * gathers the bin's residue contributions stage by stage from a small (stage, partition) -> op index built
//     in LDS (per element only the stage order matters; the accumulator starts at +0.0f like
//     the cleared buffer, so the sum is bit-identical),
//   * applies the inverse coupling steps on its registers,
//   * multiplies by the floor value of the bin (rendered once per channel into LDS by the floor phase),
//   * stores the result straight to the work plane.
// Two workgroup barriers per frame instead of eight, no spectrum round trip through LDS.
// Output contract identical to k_spectrum: work[frame][ch][0, n/2).
#include <hip/hip_runtime.h>

#include "kernels_common.h"

#ifndef S2_THREADS
#define S2_THREADS 128
#endif
#define S2_NOOP 0xFFFFu

namespace {

__constant__ float k2_inverse_db[256] = {
#include "floor1_db_table.inc"
};

__device__ __forceinline__ void s2_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ int s2_render_point(int x0, int y0, int x1, int y1, int X) {  // Floor1.cs:299-314
  int dy = y1 - y0;
  int adx = x1 - x0;
  int ady = dy < 0 ? -dy : dy;
  int err = ady * (X - x0);
  int off = err / adx;
  return dy < 0 ? y0 - off : y0 + off;
}

struct S2Floor {  // per-channel floor scratch (same content as FloorScratch of kernels_spectrum.hip)
  int fy[NVH_MAX_POSTS];
  int step[NVH_MAX_POSTS];
  int x[NVH_MAX_POSTS + 2];
  int y[NVH_MAX_POSTS + 2];
  int b[NVH_MAX_POSTS + 2];
  int ady[NVH_MAX_POSTS + 2];
  int adx[NVH_MAX_POSTS + 2];
  int nseg;
  int mode;  // 0 = channel does not execute (raw residue), 1 = floor1 curve, 2 = cleared, 3 = floor0
};

}  // namespace

// LDS map (4-byte words): [ db 256 | (coeff 256) | S2Floor x NCH | books nbooks*4 | ops cap_ops*2 | entries cap_ent/2 |
//                           opidx cap_idx/2 | ycurve NCH*block1/8 | (mult NCH*block1/2) ]   (..) = Floor0 variants only
template <int NCH, bool FLOOR0>
__device__ __forceinline__ void spectrum2_body(const NvhDevSetup& S, const NvhDevBatch& Bt, float* __restrict__ work,
                                               int* __restrict__ err, int cap_ops, int cap_ent, int cap_idx, float* smem) {
  float* s_db = smem;
  float* s_coeff = smem + 256;  // FLOOR0 only
  constexpr int HEAD = FLOOR0 ? 512 : 256;
  S2Floor* fs = reinterpret_cast<S2Floor*>(smem + HEAD);
  static_assert(sizeof(S2Floor) % 16 == 0, "alignment of what follows");
  NvhDevBook* s_books = reinterpret_cast<NvhDevBook*>(smem + HEAD + NCH * (sizeof(S2Floor) / 4));
  NvhResOp* s_ops = reinterpret_cast<NvhResOp*>(reinterpret_cast<float*>(s_books) + S.nbooks * 8);
  uint16_t* s_ent = reinterpret_cast<uint16_t*>(reinterpret_cast<float*>(s_ops) + cap_ops * 2);
  uint16_t* s_idx = s_ent + cap_ent;
  // floor curves: Floor1 keeps the inverse_dB_table index of every bin (one byte), Floor0 a float multiplier
  uint8_t* ycurve = reinterpret_cast<uint8_t*>(s_idx + cap_idx);                       // [NCH][block1/2]
  float* mult = reinterpret_cast<float*>(ycurve + NCH * (S.block1 >> 1));               // [NCH][block1/2], FLOOR0 only

  const int f = blockIdx.x;
  const NvhFrame fr = Bt.frames[f];
  if (fr.n == 0) return;
  const int half = fr.n >> 1;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const NvhChan* chans = Bt.chans + fr.chan_off;

  // ---- residue geometry of the frame's single pass (uniform) ----
  const bool has_pass = fr.pass_end > fr.pass_begin;
  NvhResPass pass;
  NvhDevResidue R;
  int nparts = 0, idx_mul = 1;
  if (has_pass) {
    pass = Bt.passes[fr.pass_begin];
    R = S.residues[pass.residue];
    // Residue0.cs:122-127 (Residue2.cs:16-21 scales the block size by the channel count first)
    const int bs = R.type == 2 ? fr.n * R.real_channels : fr.n;
    const int end = R.end < bs / 2 ? R.end : bs / 2;
    const int nn = end - R.begin;
    nparts = nn > 0 ? nn / R.partition_size : 0;
    idx_mul = R.type == 2 ? 1 : NCH;  // type 1: one op per (partition, channel)
  }

  // ---- stage side information, issue the floor loads ----
  for (int i = tid; i < 256; i += S2_THREADS) s_db[i] = k2_inverse_db[i];
  for (int i = tid; i < S.nbooks; i += S2_THREADS) s_books[i] = S.books[i];
  const bool staged = (int)fr.op_count <= cap_ops && (int)fr.ent_count <= cap_ent;
  if (staged) {
    const uint2* go = reinterpret_cast<const uint2*>(Bt.ops + fr.op_begin);
    for (int i = tid; i < (int)fr.op_count; i += S2_THREADS) reinterpret_cast<uint2*>(s_ops)[i] = go[i];
    const uint16_t* ge = Bt.entries + fr.ent_begin;
    for (int i = tid; i < (int)fr.ent_count; i += S2_THREADS) s_ent[i] = ge[i];
  }
  const int nidx = NVH_MAX_STAGES * nparts * idx_mul;
  for (int i = tid; i < nidx; i += S2_THREADS) s_idx[i] = (uint16_t)S2_NOOP;

  // ---- floor preparation: wavefront w < NCH unwraps channel w (wave-local, overlaps with the loads above) ----
  if (wv < NCH) {
    const int c = wv;
    const NvhChan chn = chans[c];
    const NvhDevFloor* fl = &S.floors[chn.floor];
    int mode = 0;
    if (chn.exec) {
      if (fl->type == 1) mode = chn.post_count > 0 ? 1 : 2;
      else mode = chn.amp > 0.0f ? 3 : 2;
    }
    if (lane == 0) fs[c].mode = mode;
    if (mode == 1) {
      const NvhDevFloor1* F = &fl->f1;
      const int pc = chn.post_count;
      const int levels = F->levels, f_range = F->range, f_mult = F->multiplier;
      int my_lo = 0, my_hi = 1, my_level = 0, my_x = 0, x_lo = 0, x_hi = 1, my_val = 0, my_sorted = 0, x_sorted = 0;
      if (lane < pc) {
        my_lo = F->l_neigh[lane];
        my_hi = F->h_neigh[lane];
        my_level = F->level[lane];
        my_x = F->x_list[lane];
        my_val = Bt.posts[chn.data_off + lane];
        my_sorted = F->sort_idx[lane];
        x_lo = F->x_list[my_lo];
        x_hi = F->x_list[my_hi];
        x_sorted = F->x_list[my_sorted];
        fs[c].fy[lane] = (lane < 2) ? my_val : 0;
        fs[c].step[lane] = (lane < 2) ? 1 : 0;
      }
      s2_wave_sync();
      // UnwrapPosts (Floor1.cs:224-297): posts of one dependency level are independent
      for (int lv = 1; lv < levels; ++lv) {
        if (lane >= 2 && lane < pc && my_level == lv) {
          int predicted = s2_render_point(x_lo, fs[c].fy[my_lo], x_hi, fs[c].fy[my_hi], my_x);
          int val = my_val;
          int highroom = f_range - predicted;
          int lowroom = predicted;
          int room = (highroom < lowroom) ? highroom * 2 : lowroom * 2;
          int fy;
          if (val != 0) {
            fs[c].step[my_lo] = 1;  // only ever set, for lower-indexed posts: order-free
            fs[c].step[my_hi] = 1;
            fs[c].step[lane] = 1;
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
          fs[c].fy[lane] = fy;
        }
        s2_wave_sync();
      }
      // Apply's walk over the sorted posts (Floor1.cs:196-216) -> compact list of line end points
      bool active = (lane < pc) && fs[c].step[my_sorted] != 0;
      unsigned long long mask = __ballot(active);
      int rank = __popcll(mask & ((1ull << lane) - 1ull));
      if (active) {
        fs[c].x[rank] = x_sorted;
        fs[c].y[rank] = fs[c].fy[my_sorted] * f_mult;
      }
      unsigned long long beyond = __ballot(active && rank >= 1 && x_sorted >= half);  // `if (lx >= n) break`
      int ns;
      if (beyond) {
        int fl0 = __ffsll((long long)beyond) - 1;
        ns = __popcll(mask & ((1ull << fl0) - 1ull));
      } else {
        ns = __popcll(mask);  // trailing flat run to n/2 (Floor1.cs:213-216)
      }
      s2_wave_sync();
      if (lane == 0) {
        if (!beyond) {
          fs[c].x[ns] = half;
          fs[c].y[ns] = fs[c].y[ns - 1];
        }
        fs[c].nseg = ns;
      }
      s2_wave_sync();
      if (lane < ns) {  // per-segment line parameters (Floor1.cs:316-326)
        int x0 = fs[c].x[lane], y0 = fs[c].y[lane];
        int x1 = fs[c].x[lane + 1] < half ? fs[c].x[lane + 1] : half;  // Math.Min(hx, n) (quirk B-6)
        int y1 = fs[c].y[lane + 1];
        int dy = y1 - y0;
        int adx = x1 - x0;
        int ady = dy < 0 ? -dy : dy;
        int b = dy / adx;
        int ab = b < 0 ? -b : b;
        fs[c].b[lane] = b;
        fs[c].ady[lane] = ady - ab * adx;
        fs[c].adx[lane] = (dy < 0) ? -adx : adx;
      }
    }
  }
  __syncthreads();  // B1: staged ops / entries / books, cleared op index, floor segments

  // ---- (stage, partition[, channel]) -> op index ----
  const NvhResOp* ops = staged ? s_ops : Bt.ops + fr.op_begin;
  const uint16_t* ent = staged ? s_ent : Bt.entries + fr.ent_begin;
  if (has_pass) {
    for (int o = tid; o < (int)fr.op_count; o += S2_THREADS) {
      const unsigned go = fr.op_begin + (unsigned)o;
      int st = 0;
#pragma unroll
      for (int k = 1; k < NVH_MAX_STAGES; ++k) st += (go >= pass.op_begin[k]) ? 1 : 0;
      const NvhResOp op = ops[o];
      if ((int)op.partition < nparts) s_idx[(st * nparts + op.partition) * idx_mul + (idx_mul > 1 ? op.channel : 0)] = (uint16_t)o;
    }
  }
  // ---- floor curves -> mult[c][x] ----
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int md = fs[c].mode;
    uint8_t* yc = ycurve + c * (S.block1 >> 1);
    float* mc = mult + c * (S.block1 >> 1);
    (void)mc;
    if (md == 1) {
      const S2Floor* Q = &fs[c];
      const int ns = Q->nseg;
      // chunks of 4 bins per thread: locate the segment, then step the reference's error-term recurrence
      // (Floor1.cs:328-340) forward, hopping segments as they end
      for (int x0 = tid * 4; x0 < half; x0 += S2_THREADS * 4) {
        int lo = 0, hi = ns - 1;
        while (lo < hi) {
          int mid = (lo + hi + 1) >> 1;
          if (Q->x[mid] <= x0) lo = mid; else hi = mid - 1;
        }
        int sg = lo;
        int sadx = Q->adx[sg], sb = Q->b[sg], sady = Q->ady[sg];
        int adx = sadx < 0 ? -sadx : sadx, sy = sadx < 0 ? -1 : 1;
        int t = x0 - Q->x[sg];
        int wq = (sady * t) / adx;
        int y = Q->y[sg] + sb * t + sy * wq;
        int e = -adx + sady * t - adx * wq;
        int xend = Q->x[sg + 1];
        unsigned packed = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int x = x0 + q;
          if (x >= xend && sg + 1 < ns) {
            ++sg;
            sadx = Q->adx[sg]; sb = Q->b[sg]; sady = Q->ady[sg];
            adx = sadx < 0 ? -sadx : sadx; sy = sadx < 0 ? -1 : 1;
            y = Q->y[sg];
            e = -adx;
            xend = Q->x[sg + 1];
          }
          int yy = y;
          if (yy < 0 || yy > 255) {
            atomicOr(err, NVH_DEVERR_FLOOR1_Y);  // inverse_dB_table[y] would throw (quirk B-7)
            yy = yy < 0 ? 0 : 255;
          }
          packed |= (unsigned)yy << (8 * q);
          y += sb;
          e += sady;
          if (e >= 0) {
            e -= adx;
            y += sy;
          }
        }
        *reinterpret_cast<unsigned*>(yc + x0) = packed;
      }
    } else if (FLOOR0 && md == 3) {  // Floor0.cs:152-212
      const NvhChan ck = chans[c];
      const NvhDevFloor0* F0 = &S.floors[ck.floor].f0;
      __syncthreads();
      for (int i = tid; i < F0->order; i += S2_THREADS) s_coeff[i] = 2.0f * (float)cos((double)Bt.coeffs[ck.data_off + i]);
      __syncthreads();
      const int slot = fr.mdct_slot;
      const int32_t* bark = S.ipool + F0->bark_off[slot];
      const float* wmap = S.fpool + F0->wmap_off[slot];
      for (int i = tid; i < half; i += S2_THREADS) {
        int kk = bark[i];
        if (kk < 0 || kk >= half) {
          atomicOr(err, NVH_DEVERR_FLOOR0_W);
          mc[i] = 0.0f;
          continue;
        }
        float p = .5f, q = .5f;
        float w = wmap[kk];
        int j;
        for (j = 1; j < F0->order; j += 2) {
          q = q * (w - s_coeff[j - 1]);
          p = p * (w - s_coeff[j]);
        }
        if (j == F0->order) {
          q = q * (w - s_coeff[j - 1]);
          p = p * (p * (4.0f - w * w));
          q = q * q;
        } else {
          p = p * (p * (2.0f - w));
          q = q * (q * (2.0f + w));
        }
        q = ck.amp / (float)sqrt((double)(p + q)) - (float)F0->amp_ofs;
        mc[i] = (float)exp((double)(q * 0.11512925f));
      }
    }
  }
  __syncthreads();  // B2: op index and floor curves complete

  // ---- per bin: gather residue (stage order), inverse coupling, floor multiply, store ----
  const NvhDevMapping mp = S.mappings[fr.mapping];
  float* planes = work + (long long)f * NCH * S.block1;
  int stage_on[NVH_MAX_STAGES];
  int nst = 0;
  if (has_pass) {
#pragma unroll
    for (int k = 0; k < NVH_MAX_STAGES; ++k)
      if (pass.op_begin[k] != pass.op_begin[k + 1]) nst = k + 1;
  }
  (void)stage_on;
  for (int x = tid; x < half; x += S2_THREADS) {
    float acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      acc[c] = 0.0f;  // Array.Clear (Mapping.cs:108)
      if (!has_pass) continue;
      // position of (bin x, channel c) inside the residue's partitioning
      int rel, cidx;
      if (R.type == 2) {
        rel = x * NCH + c - R.begin;  // interleaved (Residue2.cs:30-44; begin and partition size are channel multiples)
        cidx = 0;
      } else {
        rel = x - R.begin;            // Residue1.cs:19-22
        cidx = c;
      }
      if (rel < 0) continue;
      const unsigned p = __umulhi((unsigned)rel, R.psize_magic);
      if ((int)p >= nparts) continue;
      const unsigned i = (unsigned)rel - p * (unsigned)R.partition_size;
      for (int st = 0; st < nst; ++st) {
        const unsigned o = s_idx[(st * nparts + (int)p) * idx_mul + cidx];
        if (o == S2_NOOP) continue;
        const NvhResOp op = ops[o];
        const NvhDevBook bk = s_books[op.book];
        const unsigned j = bk.dim > 1 ? __umulhi(i, bk.dim_magic) : i;
        const unsigned comp = i - j * bk.dim;
        const unsigned e = ent[op.ent_off - fr.ent_begin + j];
        if (e == NVH_ENTRY_SKIP) continue;
        acc[c] = acc[c] + S.vq[bk.tab_off + e * bk.dim + comp];
      }
    }
    // inverse coupling, last step first (Mapping.cs:137-182)
    if (NCH == 2) {
      for (int st = mp.coupling_steps - 1; st >= 0; --st) {
        const int mg = S.coupling[mp.coupling_off + 2 * st], an = S.coupling[mp.coupling_off + 2 * st + 1];
        if (((fr.exec_mask >> an) | (fr.exec_mask >> mg)) & 1) {
          float oldM = mg == 0 ? acc[0] : acc[NCH - 1], oldA = an == 0 ? acc[0] : acc[NCH - 1], newM, newA;
          if (oldM > 0) {
            if (oldA > 0) { newM = oldM; newA = oldM - oldA; }
            else          { newA = oldM; newM = oldM + oldA; }
          } else {
            if (oldA > 0) { newM = oldM; newA = oldM + oldA; }
            else          { newA = oldM; newM = oldM - oldA; }
          }
          if (mg == 0) { acc[0] = newM; acc[NCH - 1] = newA; } else { acc[NCH - 1] = newM; acc[0] = newA; }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int md = fs[c].mode;
      float v = acc[c];
      if (md == 2) v = 0.0f;                                                    // Floor1.cs:218-221 / Floor0.cs:208-211
      else if (md == 1) v = v * s_db[ycurve[c * (S.block1 >> 1) + x]];         // v[x] *= inverse_dB_table[y]
      else if (FLOOR0 && md == 3) v = v * mult[c * (S.block1 >> 1) + x];       // residue[i] *= q
      planes[(long long)c * S.block1 + x] = v;
    }
  }
}

#define S2_KERNEL(NAME, NCH, F0)                                                                                      \
  extern "C" __global__ void __launch_bounds__(S2_THREADS)                                                            \
  NAME(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err, int cap_ops, int cap_ent,     \
       int cap_idx) {                                                                                                 \
    extern __shared__ __attribute__((aligned(16))) float smem[];                                                      \
    spectrum2_body<NCH, F0>(S, Bt, work, err, cap_ops, cap_ent, cap_idx, smem);                                       \
  }

S2_KERNEL(k_spectrum2_c1, 1, false)
S2_KERNEL(k_spectrum2_c2, 2, false)
S2_KERNEL(k_spectrum2_c1_f0, 1, true)
S2_KERNEL(k_spectrum2_c2_f0, 2, true)
