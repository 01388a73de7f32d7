// spectrum_dev.h -- device helpers shared by the spectrum kernels (kernels_spectrum.hip) and the run kernel
// (kernels_run.hip): Floor1 unwrap / segment list / curve walk (Floor1.cs:186-341), Floor0 curve (Floor0.cs:152-212),
// inverse coupling (Mapping.cs:150-178), residue element forms (Residue0.cs:180-201, Residue1.cs:8-26, Residue2.cs:23-47).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_common.h"

#define SP_THREADS 256
#define SP_GROUP 4  // channels whose floors are prepared concurrently (one wavefront each)
#ifndef SP_TAIL_BINS
#define SP_TAIL_BINS 4  // bins per lane in the fused tail
#endif

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

// One line segment of a rendered Floor1 curve: from (x, y) towards the next flagged post, drawn up to xend.
struct FloorSeg {
  uint32_t x_xend;   // x | xend << 16  (xend = start of the next segment, or n/2 / the first post beyond it)
  int32_t y;         // curve value at x (post value * multiplier)
  int32_t b;         // dy / adx                                   (Floor1.cs:316-326)
  uint32_t ady_adx;  // (|dy| - |b|*adx) | (dy < 0 ? -adx : adx) << 16
};

// Per-channel floor scratch.  The unwrap state (fy, step) is dead once the flagged posts have been compacted
// into segments, so both views share the block.
struct FloorScratch {
  union {
    struct {
      int fy[NVH_MAX_POSTS + 2];
      int step[NVH_MAX_POSTS + 2];
    } u;
    FloorSeg seg[NVH_MAX_POSTS + 2];
  };
  uint32_t magic[NVH_MAX_POSTS + 2];  // floor((2^32 - 1) / adx) of each segment: restart of the error recurrence
  int nseg;
  int mode;  // 0 = skip, 1 = floor1 curve, 2 = clear (exec without energy), 3 = floor0
};
static_assert(sizeof(FloorScratch) % 16 == 0, "keep the LDS map 16-byte aligned");
static_assert(sizeof(FloorScratch) == NVH_SP_FLOOR_SCRATCH_WORDS * 4, "host-side LDS sizing (nvh_api.hip) follows this");

// floor(n / d) for 0 <= n <= 2^31, d <= 2^16 from m = floor((2^32 - 1) / d): the estimate is at most one short
// (n*m/2^32 > n/d - (n/2^32)(1 + 1/d) >= n/d - 1).
__device__ __forceinline__ unsigned sp_div_magic(unsigned n, unsigned d, unsigned m) {
  unsigned q = __umulhi(n, m);
  return (n - q * d >= d) ? q + 1 : q;
}

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
  unsigned adx_magic;
};

__device__ __forceinline__ FloorLane load_floor_lane(const NvhDevSetup& S, const NvhDevBatch& Bt, const NvhChan* chans, int c,
                                                     int nch, int lane) {
  FloorLane L;
  L.mode = 0; L.pc = 0; L.levels = 0; L.level = 0; L.lo = 0; L.hi = 1; L.x = 0; L.x_lo = 0; L.x_hi = 1; L.val = 0;
  L.sorted = 0; L.x_sorted = 0; L.range = 0; L.mult = 0; L.adx_magic = 0;
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
    L.x_lo = F->x_lo[lane];
    L.x_hi = F->x_hi[lane];
    L.x_sorted = F->x_sorted[lane];
    L.adx_magic = F->adx_magic[lane];
  }
  return L;
}

// One wavefront turns the posts of one channel into the segment list of its curve.
//   UnwrapPosts (Floor1.cs:224-297): lane i owns post i; posts of one dependency level are independent.
//   Apply's walk over the sorted posts (Floor1.cs:196-216): the flagged posts compacted in X order; the walk
//   stops at the first end point at or beyond n/2, else a flat run to n/2 closes the curve (:213-216).
//   DUAL: one wavefront, two channels -- lanes 0..31 take channel 0 (scratch Q[0]), lanes 32..63 channel 1 (Q[1]); every
//   post count involved is <= 32 (the caller checks).  The unwrap is a chain of dependent LDS round trips with at most
//   `posts` lanes busy, so two channels side by side cost the time of one.
template <bool DUAL = false>
__device__ __forceinline__ void floor_prepare(FloorScratch* Q, const FloorLane& L, int lane, int half, int* __restrict__ err,
                                              const uint32_t* __restrict__ recip) {
  const int mode = L.mode, pc = L.pc;
  const int wlane = lane;  // position in the wavefront (ballot masks)
  if (DUAL) {
    Q += lane >> 5;
    lane &= 31;
  }
  const bool on = mode == 1;  // uniform over the wavefront (DUAL: over its half)
  if (lane == 0) Q->mode = mode;
  if (!DUAL && !on) return;
  int levels = L.levels;
  if (DUAL) {
    const int l0 = __shfl(on ? L.levels : 0, 0), l1 = __shfl(on ? L.levels : 0, 32);
    levels = l0 > l1 ? l0 : l1;
    if (levels == 0) return;  // neither channel draws a curve
  }
  // the part of a 64-lane ballot that belongs to this lane's channel
  auto mine = [&](unsigned long long m) -> unsigned long long { return DUAL ? ((m >> (wlane & 32)) & 0xFFFFFFFFull) : m; };
  if (on && lane < pc) {
    Q->u.fy[lane] = (lane < 2) ? L.val : 0;
    Q->u.step[lane] = (lane < 2) ? 1 : 0;
  }
  sp_wave_sync();
  for (int lv = 1; lv < levels; ++lv) {
    if (on && lane >= 2 && lane < pc && L.level == lv) {
      // RenderPoint (Floor1.cs:299-314) with the static divisor's reciprocal
      int predicted;
      {
        const int y0 = Q->u.fy[L.lo], y1 = Q->u.fy[L.hi];
        const int dy = y1 - y0, adx = L.x_hi - L.x_lo;
        const int ady = dy < 0 ? -dy : dy;
        const int er = (int)((unsigned)ady * (unsigned)(L.x - L.x_lo));  // the managed product wraps (unchecked int)
        // truncating division of the (possibly wrapped, hence negative) product: |er| <= 2^31 and adx <= 2^13 keep
        // the reciprocal estimate within one of the quotient
        const unsigned aer = er < 0 ? 0u - (unsigned)er : (unsigned)er;
        const int qa = (int)sp_div_magic(aer, (unsigned)adx, L.adx_magic);
        const int off = er < 0 ? -qa : qa;
        predicted = dy < 0 ? y0 - off : y0 + off;
      }
      int val = L.val;
      int highroom = L.range - predicted;
      int lowroom = predicted;
      int room = (highroom < lowroom) ? highroom * 2 : lowroom * 2;
      int fy;
      if (val != 0) {
        // stepFlags are only ever set, never cleared: order-free
        Q->u.step[L.lo] = 1;
        Q->u.step[L.hi] = 1;
        Q->u.step[lane] = 1;
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
      Q->u.fy[lane] = fy;
    }
    sp_wave_sync();
  }
  // compact the flagged posts in X order; the unwrap state is read into registers before the segment view
  // (which shares its storage) is written
  const bool active = on && (lane < pc) && Q->u.step[L.sorted] != 0;
  const int ys = (on && lane < pc) ? Q->u.fy[L.sorted] * L.mult : 0;
  const unsigned long long mask = mine(__ballot(active));
  const int rank = __popcll(mask & ((1ull << lane) - 1ull));
  const unsigned long long beyond = mine(__ballot(active && rank >= 1 && L.x_sorted >= half));
  int ns;
  if (beyond) {
    const int fl0 = __ffsll((long long)beyond) - 1;
    ns = __popcll(mask & ((1ull << fl0) - 1ull));
  } else {
    ns = __popcll(mask);  // trailing flat run to n/2
  }
  if (!on) ns = 0;
  sp_wave_sync();
  if (active) {
    Q->seg[rank].x_xend = (uint32_t)L.x_sorted;
    Q->seg[rank].y = ys;
  }
  sp_wave_sync();
  if (on && lane == 0) {
    if (!beyond) {
      Q->seg[ns].x_xend = (uint32_t)half;
      Q->seg[ns].y = Q->seg[ns - 1].y;
    }
    Q->nseg = ns;
  }
  sp_wave_sync();
  int x0 = 0, x1n = 0, y0 = 0, y1 = 0;
  if (lane < ns) {
    x0 = (int)Q->seg[lane].x_xend;
    y0 = Q->seg[lane].y;
    x1n = (int)Q->seg[lane + 1].x_xend;
    y1 = Q->seg[lane + 1].y;
  }
  sp_wave_sync();  // every lane has read its successor's plain x before the packed form goes in
  if (lane < ns) {
    const int x1 = x1n < half ? x1n : half;  // Math.Min(hx, n) (quirk B-6)
    const int dy = y1 - y0;
    const int adx = x1 - x0;
    const int ady = dy < 0 ? -dy : dy;
    // floor((2^32 - 1) / adx) from the setup's table (1 <= adx <= n/2): it restarts the error recurrence in the tail and
    // gives b = dy / adx (truncating) here -- two emulated 32-bit divisions per segment otherwise
    const unsigned mg = recip[adx];
    const int ab = (int)sp_div_magic((unsigned)ady, (unsigned)adx, mg);
    const int b = dy < 0 ? -ab : ab;
    const int ady2 = ady - ab * adx;
    FloorSeg sgm;
    sgm.x_xend = (uint32_t)x0 | ((uint32_t)x1n << 16);
    sgm.y = y0;
    sgm.b = b;
    sgm.ady_adx = ((uint32_t)ady2 & 0xFFFFu) | ((uint32_t)((dy < 0) ? -adx : adx) << 16);
    Q->seg[lane] = sgm;
    Q->magic[lane] = mg;
    // inverse_dB_table[y] throws for y outside 0..255 (quirk B-7).  The curve is monotone inside a segment, so
    // its first and last drawn values decide; the render loop itself then only clamps.
    const int tl = adx - 1;
    const int yl = y0 + b * tl + (dy < 0 ? -1 : 1) * (int)sp_div_magic((unsigned)(ady2 * tl), (unsigned)adx, mg);
    if (y0 < 0 || y0 > 255 || yl < 0 || yl > 255) atomicOr(err, NVH_DEVERR_FLOOR1_Y);
  }
}

// Curve values of NB consecutive bins starting at x0 (a multiple of NB): locate the segment once, restart the
// reference's error-term recurrence (Floor1.cs:328-340) from its closed form, then step it, hopping segments
// as they end.  Returns the NB inverse-dB multipliers.
template <int NB>
__device__ __forceinline__ void floor_walk_seg(const FloorSeg* __restrict__ seg, const uint32_t* __restrict__ magic, int ns,
                                               const float* __restrict__ s_db, int x0, float m[NB],
                                               const uint8_t* __restrict__ segtab = nullptr) {
  // last segment whose start is <= x0: from the per-four-bins table when the caller has one, else a
  // fixed-trip binary search (the trip count depends on ns only, so the loop control is scalar; the data-dependent form
  // costs an exec-mask loop per lane)
  int sg = 0;
  if (segtab) {
    sg = segtab[x0 >> 2];
  } else {
#pragma unroll
    for (int step = 64; step > 0; step >>= 1) {
      if (step >= ns) continue;  // wave-uniform
      const int cand = sg + step;
      const int ci = cand < ns ? cand : ns - 1;
      const int xs = (int)(seg[ci].x_xend & 0xFFFFu);
      if (cand < ns && xs <= x0) sg = cand;
    }
  }
  FloorSeg s = seg[sg];
  int sadx = (int)s.ady_adx >> 16, sady = (int)(s.ady_adx & 0xFFFFu), sb = s.b;
  int adx = sadx < 0 ? -sadx : sadx, sy = sadx < 0 ? -1 : 1;
  const int t = x0 - (int)(s.x_xend & 0xFFFFu);
  const int wq = (int)sp_div_magic((unsigned)(sady * t), (unsigned)adx, magic[sg]);  // sady, t < adx <= 2^13
  int y = s.y + sb * t + sy * wq;
  int e = -adx + sady * t - adx * wq;  // the reference's `err` after t steps
  int xend = (int)(s.x_xend >> 16);
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int x = x0 + q;
    // the next segment starts exactly here; the last segment ends at or beyond n/2 (floor_prepare: the walk stops at the
    // first end point >= n/2, else the closing flat run ends at n/2), so x < n/2 never runs off the list
    if (x >= xend) {
      ++sg;
      s = seg[sg];
      sadx = (int)s.ady_adx >> 16; sady = (int)(s.ady_adx & 0xFFFFu); sb = s.b;
      adx = sadx < 0 ? -sadx : sadx; sy = sadx < 0 ? -1 : 1;
      y = s.y;
      e = -adx;
      xend = (int)(s.x_xend >> 16);
    }
    const int yy = y < 0 ? 0 : (y > 255 ? 255 : y);  // out-of-range values were reported by floor_prepare
    m[q] = s_db[yy];
    y += sb;  // advance to x+1 inside the segment
    e += sady;
    if (e >= 0) {
      e -= adx;
      y += sy;
    }
  }
}

template <int NB>
__device__ __forceinline__ void floor_walk(const FloorScratch* Q, const float* __restrict__ s_db, int x0, float m[NB]) {
  floor_walk_seg<NB>(Q->seg, Q->magic, __builtin_amdgcn_readfirstlane(Q->nseg), s_db, x0, m);
}

// Floor0.Apply's curve (Floor0.cs:152-212) for one channel with Amp > 0: all NT threads of the workgroup; s_coeff holds
// `order` floats.  The reference evaluates once per run of equal barkMap entries and reuses the value: the same
// arithmetic per bin here.
template <int NT>
__device__ __forceinline__ void floor0_curve(const NvhDevSetup& S, const NvhDevFloor0* F0, const float* __restrict__ coeff,
                                             float amp, int slot, float* res, int half, float* s_coeff, int tid,
                                             int* __restrict__ err) {
  __syncthreads();
  for (int i = tid; i < F0->order; i += NT) s_coeff[i] = 2.0f * (float)cos((double)coeff[i]);
  __syncthreads();
  const int32_t* bark = S.ipool + F0->bark_off[slot];
  const float* wmap = S.fpool + F0->wmap_off[slot];
  for (int i = tid; i < half; i += NT) {
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
    q = amp / (float)sqrt((double)(p + q)) - (float)F0->amp_ofs;
    q = (float)exp((double)(q * 0.11512925f));
    res[i] = res[i] * q;
  }
}

// Mapping.cs:150-178 without branches.  The reference's four cases
//   M > 0, A > 0: (M, M - A)    M > 0, A <= 0: (M + A, M)    M <= 0, A > 0: (M, M + A)    M <= 0, A <= 0: (M - A, M)
// all compute v = M +/- A, subtracting exactly when the two comparisons agree, and put v in the angle slot when A > 0,
// in the magnitude slot otherwise.  M - A and M + (-A) are the same IEEE operation.
__device__ __forceinline__ void couple1(float& M, float& A) {
  const float oldM = M, oldA = A;
  const bool mpos = oldM > 0, apos = oldA > 0;
  const float v = oldM + ((mpos == apos) ? -oldA : oldA);
  M = apos ? oldM : v;
  A = apos ? v : oldM;
}

}  // namespace
