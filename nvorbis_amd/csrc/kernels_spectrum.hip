// kernels_spectrum.hip -- fused spectrum kernel: residue vector adds + inverse coupling + floor apply with the
// whole frame's spectrum resident in LDS (one workgroup per frame, all channels).
//
//   Array.Clear + IResidue.Decode adds   Mapping.cs:108,133; Residue0.cs:180-201, Residue1.cs:8-26, Residue2.cs:23-47
//   inverse square-polar coupling         Mapping.cs:137-182
//   IFloor.Apply                          Floor1.cs:186-341 (UnwrapPosts :224-297), Floor0.cs:152-212
//
// Output: work[frame][ch][0, n/2) = the vector IMdct.Reverse consumes (or, for a channel that does not
// execute, the raw residue -- quirk B-4).  The IMDCT kernel (kernels_imdct.hip) picks it up from there.
//
// The kernel is latency-, not bandwidth-bound (a frame's side information is ~2 KB), so its structure is
// about short dependency chains: the frame's op list, entry stream and the codebook directory are staged
// into LDS with one coalesced burst, every lane of the floor unwrap fetches its static post geometry up
// front, and all index divisions are exact reciprocal multiplies prepared by the host.
// Bit-exactness: residue adds replay the reference's stage order (one barrier per stage); all float
// expressions are single operations; -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "kernels_common.h"

#define SP_THREADS 256
#define SP_GROUP 4  // channels whose floors are prepared concurrently (one wavefront each)

namespace {

__constant__ float k_inverse_db[256] = {
#include "floor1_db_table.inc"
};

__device__ __forceinline__ int sp_render_point(int x0, int y0, int x1, int y1, int X) {  // Floor1.cs:299-314
  int dy = y1 - y0;
  int adx = x1 - x0;
  int ady = dy < 0 ? -dy : dy;
  int err = ady * (X - x0);
  int off = err / adx;
  return dy < 0 ? y0 - off : y0 + off;
}

// A wavefront's floor scratch block is private to it (wave w prepares channel c0 + w): LDS ordering inside the
// wave only needs the compiler to keep program order.
__device__ __forceinline__ void sp_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct FloorScratch {
  int fy[NVH_MAX_POSTS];
  int step[NVH_MAX_POSTS];
  // line segments k = 0 .. nseg-1: from (x[k], y[k]) towards (x[k+1], y[k+1]), drawn up to min(x[k+1], n/2)
  int x[NVH_MAX_POSTS + 2];
  int y[NVH_MAX_POSTS + 2];
  int b[NVH_MAX_POSTS + 2];     // dy / adx
  int ady[NVH_MAX_POSTS + 2];   // |dy| - |b|*adx
  int adx[NVH_MAX_POSTS + 2];   // x1 - x0, negated when dy < 0
  int nseg;
  int mode;  // 0 = skip, 1 = floor1 curve, 2 = clear (exec without energy), 3 = floor0
};

// General (division-based) form of one residue element, all residue types.
__device__ __forceinline__ void residue_apply_lds(const NvhDevBook bk, const float* __restrict__ vq, const NvhDevResidue& R,
                                                  const NvhResOp op, const uint16_t* __restrict__ ent, unsigned ent_begin, int i,
                                                  float* spec, int half) {
  const int dims = (int)bk.dim;
  const int offset = R.begin + (int)op.partition * R.partition_size;
  int j, comp, ch, x;
  if (R.type == 0) {
    int steps = R.partition_size / dims;  // Residue0.cs:183,193-199: res[offset++] over dim-major order
    if (i >= steps * dims) return;
    comp = i / steps;
    j = i - comp * steps;
    ch = op.channel;
    x = offset + i;
  } else if (R.type == 1) {  // Residue1.cs:19-22
    j = i / dims;
    comp = i - j * dims;
    ch = op.channel;
    x = offset + i;
  } else {  // Residue2.cs:25-45: offset /= channels; chPtr restarts at 0 (quirk B-1)
    j = i / dims;
    comp = i - j * dims;
    ch = i % R.real_channels;
    x = offset / R.real_channels + i / R.real_channels;
  }
  unsigned e = ent[op.ent_off - ent_begin + j];
  if (e == NVH_ENTRY_SKIP) return;
  if (x >= half) return;  // lands in [n/2, block1): overwritten by the IMDCT or cleared, never observed
  float* p = spec + ch * half + x;
  *p = *p + vq[bk.tab_off + e * (unsigned)dims + (unsigned)comp];
}

// Residue types 1 and 2 with every division replaced by an exact reciprocal multiply (NvhDevResidue::fast).
// Split in two so that the caller can have several independent element chains (op -> book -> entry -> value) in
// flight before it commits the adds: returns the target (nullptr: nothing to add) and the value.
__device__ __forceinline__ float* residue_fetch_fast(const NvhDevBook* __restrict__ s_books, const float* __restrict__ vq,
                                                     const NvhDevResidue& R, const NvhResOp op,
                                                     const uint16_t* __restrict__ ent, unsigned ent_begin, int i, float* spec,
                                                     int half, const uint32_t* __restrict__ s_lat, float* val) {
  const NvhDevBook bk = s_books[op.book];
  const unsigned dims = bk.dim;
  const unsigned j = dims > 1 ? __umulhi((unsigned)i, bk.dim_magic) : (unsigned)i;
  const unsigned comp = (unsigned)i - j * dims;
  const unsigned e = ent[op.ent_off - ent_begin + j];
  if (e == NVH_ENTRY_SKIP) return nullptr;
  const int offset = R.begin + (int)op.partition * R.partition_size;
  int ch, x;
  if (R.type == 1) {
    ch = op.channel;
    x = offset + i;
  } else {
    const unsigned rch = (unsigned)R.real_channels;
    if (rch > 1) {
      const unsigned qi = __umulhi((unsigned)i, R.rch_magic);
      ch = (int)((unsigned)i - qi * rch);
      x = (int)(__umulhi((unsigned)offset, R.rch_magic) + qi);
    } else {
      ch = 0;
      x = offset + i;
    }
  }
  if (x >= half) return nullptr;
  if (bk.lat_values) {
    // lattice book: component = distinct[(e / lat_values^comp) % lat_values], all in LDS, no table gather
    const uint32_t pm = s_lat[bk.lat_off + bk.lat_values + comp];
    const unsigned q = pm ? __umulhi(e, pm) : e;
    const unsigned digit = bk.lat_values > 1 ? q - __umulhi(q, bk.lat_magic) * bk.lat_values : 0u;
    *val = __uint_as_float(s_lat[bk.lat_off + digit]);
  } else {
    *val = vq[bk.tab_off + e * dims + comp];
  }
  return spec + ch * half + x;
}

// Everything lane i of a wavefront needs to unwrap post i of its channel: fetched with independent loads so that
// one memory latency covers the lot (static post geometry from the setup, the raw post value from the batch).
struct FloorLane {
  int mode;  // 0 skip, 1 floor1 curve, 2 clear, 3 floor0
  int pc, levels, level, lo, hi, x, x_lo, x_hi, val, sorted, x_sorted, range, mult;
};

__device__ __forceinline__ FloorLane load_floor_lane(const NvhDevSetup& S, const NvhDevBatch& Bt, const NvhChan* chans, int c,
                                                     int nch, int lane) {
  FloorLane L;
  L.mode = 0; L.pc = 0; L.levels = 0; L.level = 0; L.lo = 0; L.hi = 1; L.x = 0; L.x_lo = 0; L.x_hi = 1; L.val = 0;
  L.sorted = 0; L.x_sorted = 0; L.range = 0; L.mult = 0;
  if (c >= nch) return L;
  const NvhChan chn = chans[c];
  const NvhDevFloor* fl = &S.floors[chn.floor];
  if (chn.exec) {
    if (fl->type == 1) L.mode = chn.post_count > 0 ? 1 : 2;
    else L.mode = chn.amp > 0.0f ? 3 : 2;
  }
  if (L.mode != 1) return L;
  const NvhDevFloor1* F = &fl->f1;
  L.pc = chn.post_count;
  L.levels = F->levels;
  L.range = F->range;
  L.mult = F->multiplier;
  if (lane < L.pc) {
    L.lo = F->l_neigh[lane];
    L.hi = F->h_neigh[lane];
    L.level = F->level[lane];
    L.x = F->x_list[lane];
    L.val = Bt.posts[chn.data_off + lane];
    L.sorted = F->sort_idx[lane];
    L.x_lo = F->x_list[L.lo];
    L.x_hi = F->x_list[L.hi];
    L.x_sorted = F->x_list[L.sorted];
  }
  return L;
}

}  // namespace

// LDS map (dynamic, 4-byte words): [ s_db 256 | s_coeff 256 | FloorScratch x SP_GROUP | books nbooks*8 | lattice pool |
//                                   ops cap_ops*2 | entries cap_ent/2 | spectrum ch*half ]
// cap_ops / cap_ent == 0: the frame's ops / entries are read from global memory instead (oversized frames).
template <bool FLOOR0>
__device__ __forceinline__ void spectrum_body(const NvhDevSetup& S, const NvhDevBatch& Bt, float* __restrict__ work,
                                              int* __restrict__ err, int phase_mask, int cap_ops, int cap_ent,
                                              float* smem, long long* dbg = nullptr) {
  float* s_db = smem;
  float* s_coeff = smem + 256;
  FloorScratch* fs = reinterpret_cast<FloorScratch*>(smem + 512);
  static_assert(sizeof(FloorScratch) % 16 == 0, "keep the spectrum 16-byte aligned");
  NvhDevBook* s_books = reinterpret_cast<NvhDevBook*>(smem + 512 + SP_GROUP * (sizeof(FloorScratch) / 4));
  uint32_t* s_lat = reinterpret_cast<uint32_t*>(reinterpret_cast<float*>(s_books) + S.nbooks * 8);
  NvhResOp* s_ops = reinterpret_cast<NvhResOp*>(s_lat + ((S.lattice_words + 3) & ~3));
  uint16_t* s_ent = reinterpret_cast<uint16_t*>(reinterpret_cast<float*>(s_ops) + cap_ops * 2);
  float* spec = reinterpret_cast<float*>(s_ent) + ((cap_ent + 7) >> 3) * 4;  // [ch][half], 16-byte aligned

  const int f = blockIdx.x;
#define DBG_T(k) do { if (dbg && threadIdx.x == 0) dbg[(long long)blockIdx.x * 24 + (k)] = clock64(); } while (0)
  DBG_T(0);
  const NvhFrame fr = Bt.frames[f];
  if (fr.n == 0) return;
  const int half = fr.n >> 1;
  const int tid = threadIdx.x;
  const int nch = S.channels;
  const NvhChan* chans = Bt.chans + fr.chan_off;
  if (dbg && fr.n == 12345) dbg[0] = 0;
  DBG_T(1);

  // floor lane data of the first channel group: independent of the residue, so fetch it now and let the
  // latency hide behind the residue and coupling phases
  const int wv = tid >> 6, lane = tid & 63;
  const FloorLane first_lane = load_floor_lane(S, Bt, chans, wv, nch, lane);

  // ---- stage the frame's side information (one coalesced burst) and clear the spectrum ----
  const bool staged = (int)fr.op_count <= cap_ops && (int)fr.ent_count <= cap_ent;
  s_db[tid] = k_inverse_db[tid];
  for (int i = tid; i < S.nbooks; i += SP_THREADS) s_books[i] = S.books[i];
  for (int i = tid; i < S.lattice_words; i += SP_THREADS) s_lat[i] = S.lattice[i];
  if (staged) {
    const uint2* go = reinterpret_cast<const uint2*>(Bt.ops + fr.op_begin);
    for (int i = tid; i < (int)fr.op_count; i += SP_THREADS) reinterpret_cast<uint2*>(s_ops)[i] = go[i];
    const uint16_t* ge = Bt.entries + fr.ent_begin;
    for (int i = tid; i < (int)fr.ent_count; i += SP_THREADS) s_ent[i] = ge[i];
  }
  for (int i = tid; i < nch * half; i += SP_THREADS) spec[i] = 0.0f;  // Mapping.cs:108
  __syncthreads();
  DBG_T(2);
  // both sources are indexed relative to the frame's slice (op.ent_off and pass->op_begin[] are batch offsets)
  const NvhResOp* ops = staged ? s_ops : Bt.ops + fr.op_begin;
  const uint16_t* ent = staged ? s_ent : Bt.entries + fr.ent_begin;

  // ---- residue ----  (phase_mask: profiling aid, all bits set in production)
  for (unsigned ps = fr.pass_begin; (phase_mask & 1) && ps < fr.pass_end; ++ps) {
    // by value: the stage loop below is full of barriers, across which loads through a pointer are not hoisted --
    // every stage (empty ones included) would pay a scalar-load round trip for its op range
    const NvhResPass pass = Bt.passes[ps];
    const NvhDevResidue R = S.residues[pass.residue];
    const int psize = R.partition_size;
    if (dbg && psize == 123456) dbg[1] = 0;
    long long t_prev = dbg ? clock64() : 0;
    if (dbg && threadIdx.x == 0) dbg[(long long)blockIdx.x * 24 + 7] = t_prev;
#pragma unroll
    for (int s = 0; s < NVH_MAX_STAGES; ++s) {
      const unsigned ob = pass.op_begin[s] - fr.op_begin, oe = pass.op_begin[s + 1] - fr.op_begin;
      if (ob == oe) continue;
      if (!R.sequential && R.fast) {
        // elements of one stage never alias (that is what !sequential means), so four of them are fetched as
        // independent dependency chains before their adds are committed
        const int total = (int)(oe - ob) * psize;
        for (int base = tid; base < total; base += 4 * SP_THREADS) {
          float* tp[4];
          float tv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int idx = base + u * SP_THREADS;
            tp[u] = nullptr;
            tv[u] = 0.0f;
            if (idx < total) {
              unsigned o = __umulhi((unsigned)idx, R.psize_magic);
              int i = idx - (int)o * psize;
              tp[u] = residue_fetch_fast(s_books, S.vq, R, ops[ob + o], ent, fr.ent_begin, i, spec, half, s_lat, &tv[u]);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (tp[u]) *tp[u] = *tp[u] + tv[u];
        }
        __syncthreads();
        if (dbg) {
          long long t_now = clock64();
          if (threadIdx.x == 0) {
            dbg[(long long)blockIdx.x * 24 + 8 + s] = t_now - t_prev;
            dbg[(long long)blockIdx.x * 24 + 16 + s] = (long long)(oe - ob);
          }
          t_prev = t_now;
        }
      } else if (!R.sequential) {
        const int total = (int)(oe - ob) * psize;
        for (int idx = tid; idx < total; idx += SP_THREADS) {
          unsigned o = (unsigned)idx / (unsigned)psize;
          int i = idx - (int)o * psize;
          const NvhResOp op = ops[ob + o];
          residue_apply_lds(s_books[op.book], S.vq, R, op, ent, fr.ent_begin, i, spec, half);
        }
        __syncthreads();
      } else {
        // partitions may alias (quirk B-1 / vector overrun): keep the reference's partition order
        for (unsigned o = ob; o < oe; ++o) {
          const NvhResOp op = ops[o];
          const NvhDevBook bk = s_books[op.book];
          const int dims = (int)bk.dim;
          const int cnt = ((psize + dims - 1) / dims) * dims;
          for (int i = tid; i < cnt; i += SP_THREADS) residue_apply_lds(bk, S.vq, R, op, ent, fr.ent_begin, i, spec, half);
          __syncthreads();
        }
      }
    }
  }

  DBG_T(3);
  // ---- inverse coupling, last step first (Mapping.cs:137-182) ----
  const NvhDevMapping mp = S.mappings[fr.mapping];
  for (int st = mp.coupling_steps - 1; (phase_mask & 2) && st >= 0; --st) {
    const int mg = S.coupling[mp.coupling_off + 2 * st], an = S.coupling[mp.coupling_off + 2 * st + 1];
    if (chans[an].exec || chans[mg].exec) {
      float* M = spec + mg * half;
      float* Aa = spec + an * half;
      for (int j = tid; j < half; j += SP_THREADS) {
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

  DBG_T(4);
  // ---- floors, SP_GROUP channels at a time: wavefront w prepares channel c0 + w ----
  for (int c0 = 0; (phase_mask & 4) && c0 < nch; c0 += SP_GROUP) {
    FloorLane fl_lane = (c0 == 0) ? first_lane : load_floor_lane(S, Bt, chans, c0 + wv, nch, lane);
    const int mode = fl_lane.mode;
    const int pc = fl_lane.pc;
    if (lane == 0) fs[wv].mode = mode;
    // UnwrapPosts (Floor1.cs:224-297).  Lane i owns post i; posts of one dependency level are independent, and the
    // scratch block belongs to this wavefront alone, so the levels are separated by wave-local ordering only.
    if (lane < pc) {
      fs[wv].fy[lane] = (lane < 2) ? fl_lane.val : 0;
      fs[wv].step[lane] = (lane < 2) ? 1 : 0;
    }
    sp_wave_sync();
    for (int lv = 1; lv < fl_lane.levels; ++lv) {
      if (lane >= 2 && lane < pc && fl_lane.level == lv) {
        int predicted = sp_render_point(fl_lane.x_lo, fs[wv].fy[fl_lane.lo], fl_lane.x_hi, fs[wv].fy[fl_lane.hi], fl_lane.x);
        int val = fl_lane.val;
        int highroom = fl_lane.range - predicted;
        int lowroom = predicted;
        int room = (highroom < lowroom) ? highroom * 2 : lowroom * 2;
        int fy;
        if (val != 0) {
          // stepFlags are only ever set for lower-indexed posts, never cleared afterwards: order-free
          fs[wv].step[fl_lane.lo] = 1;
          fs[wv].step[fl_lane.hi] = 1;
          fs[wv].step[lane] = 1;
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
        fs[wv].fy[lane] = fy;
      }
      sp_wave_sync();
    }
    const int my_sorted = fl_lane.sorted, x_sorted = fl_lane.x_sorted, f_mult = fl_lane.mult;
    // Apply's walk over the sorted posts (Floor1.cs:196-216): compact the flagged posts in X order
    if (mode == 1) {
      bool active = (lane < pc) && fs[wv].step[my_sorted] != 0;
      unsigned long long mask = __ballot(active);
      int rank = __popcll(mask & ((1ull << lane) - 1ull));
      if (active) {
        fs[wv].x[rank] = x_sorted;
        fs[wv].y[rank] = fs[wv].fy[my_sorted] * f_mult;
      }
      // the walk stops at the first end point at or beyond n/2 (`if (lx >= n) break`)
      unsigned long long beyond = __ballot(active && rank >= 1 && x_sorted >= half);
      int nact = __popcll(mask);
      int ns;
      if (beyond) {
        int fl0 = __ffsll((long long)beyond) - 1;
        ns = __popcll(mask & ((1ull << fl0) - 1ull));
      } else {
        ns = nact;  // trailing flat run to n/2 (Floor1.cs:213-216)
      }
      sp_wave_sync();
      if (lane == 0) {
        if (!beyond) {
          fs[wv].x[ns] = half;
          fs[wv].y[ns] = fs[wv].y[ns - 1];
        }
        fs[wv].nseg = ns;
      }
      sp_wave_sync();
      // per-segment line parameters (Floor1.cs:316-326): one lane per segment
      if (lane < ns) {
        int x0 = fs[wv].x[lane], y0 = fs[wv].y[lane];
        int x1 = fs[wv].x[lane + 1] < half ? fs[wv].x[lane + 1] : half;  // Math.Min(hx, n) (quirk B-6)
        int y1 = fs[wv].y[lane + 1];
        int dy = y1 - y0;
        int adx = x1 - x0;
        int ady = dy < 0 ? -dy : dy;
        int b = dy / adx;
        int ab = b < 0 ? -b : b;
        fs[wv].b[lane] = b;
        fs[wv].ady[lane] = ady - ab * adx;
        fs[wv].adx[lane] = (dy < 0) ? -adx : adx;
      }
    }
    __syncthreads();

    // ---- render / apply: all threads over the group's channels ----
    const int ngrp = (nch - c0) < SP_GROUP ? (nch - c0) : SP_GROUP;
    for (int k = 0; k < ngrp; ++k) {
      const int cc = c0 + k;
      const int md = fs[k].mode;
      float* res = spec + cc * half;
      if (md == 0) continue;
      if (md == 2) {
        for (int i = tid; i < half; i += SP_THREADS) res[i] = 0.0f;  // Floor1.cs:218-221 / Floor0.cs:208-211
        continue;
      }
      if (md == 1) {
        const FloorScratch* Q = &fs[k];
        const int ns = Q->nseg;
        // chunks of 4 consecutive bins per thread: locate the segment once, then step the reference's
        // error-term recurrence (Floor1.cs:328-340) forward, hopping segments as they end
        for (int x0 = tid * 4; x0 < half; x0 += SP_THREADS * 4) {
          int lo = 0, hi = ns - 1;
          while (lo < hi) {  // last segment whose start is <= x0
            int mid = (lo + hi + 1) >> 1;
            if (Q->x[mid] <= x0) lo = mid; else hi = mid - 1;
          }
          int sg = lo;
          int sx = Q->x[sg], sadx = Q->adx[sg], sb = Q->b[sg], sady = Q->ady[sg];
          int adx = sadx < 0 ? -sadx : sadx, sy = sadx < 0 ? -1 : 1;
          int t = x0 - sx;
          int wq = (sady * t) / adx;
          int y = Q->y[sg] + sb * t + sy * wq;
          int e = -adx + sady * t - adx * wq;  // the reference's `err` after t steps
          int xend = Q->x[sg + 1];
          float4 v = *reinterpret_cast<float4*>(res + x0);
          float m[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            int x = x0 + q;
            if (x >= xend && sg + 1 < ns) {  // the next segment starts exactly here
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
            m[q] = s_db[yy];
            y += sb;  // advance to x+1 inside the segment
            e += sady;
            if (e >= 0) {
              e -= adx;
              y += sy;
            }
          }
          v.x = v.x * m[0];
          v.y = v.y * m[1];
          v.z = v.z * m[2];
          v.w = v.w * m[3];
          *reinterpret_cast<float4*>(res + x0) = v;
        }
        continue;
      }
      if (FLOOR0) {  // Floor0 (Floor0.cs:152-212)
        const NvhChan ck = chans[cc];
        const NvhDevFloor0* F0 = &S.floors[ck.floor].f0;
        __syncthreads();
        for (int i = tid; i < F0->order; i += SP_THREADS) s_coeff[i] = 2.0f * (float)cos((double)Bt.coeffs[ck.data_off + i]);
        __syncthreads();
        const int slot = fr.mdct_slot;
        const int32_t* bark = S.ipool + F0->bark_off[slot];
        const float* wmap = S.fpool + F0->wmap_off[slot];
        for (int i = tid; i < half; i += SP_THREADS) {
          int kk = bark[i];
          if (kk < 0 || kk >= half) {
            atomicOr(err, NVH_DEVERR_FLOOR0_W);
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
          q = (float)exp((double)(q * 0.11512925f));
          res[i] = res[i] * q;
        }
      }
    }
    __syncthreads();
  }

  DBG_T(5);
  // ---- spectrum -> work planes ----
  float* planes = work + (long long)f * nch * S.block1;
  const int q4 = half >> 2;
  for (int i = tid; i < nch * q4; i += SP_THREADS) {
    int c = i / q4, k = i - c * q4;
    reinterpret_cast<float4*>(planes + (long long)c * S.block1)[k] = reinterpret_cast<const float4*>(spec + c * half)[k];
  }
  DBG_T(6);
}

extern "C" __global__ void __launch_bounds__(SP_THREADS)
k_spectrum(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err, int phase_mask, int cap_ops,
           int cap_ent, long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  spectrum_body<false>(S, Bt, work, err, phase_mask, cap_ops, cap_ent, smem, dbg);
}

// Variant for setups that contain a Floor0 (double-precision cos / sqrt / exp: costs registers, kept apart).
extern "C" __global__ void __launch_bounds__(SP_THREADS)
k_spectrum_f0(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err, int phase_mask, int cap_ops,
              int cap_ent) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  spectrum_body<true>(S, Bt, work, err, phase_mask, cap_ops, cap_ent, smem);
}
