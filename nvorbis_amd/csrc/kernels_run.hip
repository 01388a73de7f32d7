// kernels_run.hip -- the run kernel: packets' side information in, interleaved clipped PCM out, in ONE launch.
//
//   Array.Clear + IResidue.Decode adds      Mapping.cs:108,133; Residue1.cs:8-26, Residue2.cs:23-47
//   inverse square-polar coupling            Mapping.cs:137-182
//   IFloor.Apply (Floor1)                    Floor1.cs:186-341
//   IMdct.Reverse                            Mdct.cs:65-313 (imdct_wave.h)
//   window multiply                          Mode.cs:160-166
//   OverlapBuffers + ReadNextPacket          StreamDecoder.cs:417-463, 532-541
//   ClippingCopyBuffer / CopyBuffer          StreamDecoder.cs:391-415, Utils.cs:30-43
//
// One workgroup decodes a RUN of consecutive frames, highest index first.  The reference overlaps the windowed tail of
// block f-1 onto the head of block f (next[start + j] += previous[prevEnd + j]); float addition is commutative, so the
// order in which the two blocks are computed does not matter.  Walking the run downwards, the head of the block computed
// last ("parked" in LDS, n/2 floats per channel) is completed by the tail of the block computed now and goes out as PCM;
// nothing of a block ever travels through HBM except at a run boundary:
//   * the first block a workgroup computes (the run's last frame) stores its windowed tail to global memory and raises a
//     flag -- at the START of the workgroup's life;
//   * the last block a workgroup computes (the run's first frame) needs the tail of the frame before it, which the
//     neighbouring workgroup published when it started, i.e. a whole run earlier -- at the END of this workgroup's life.
// So the one cross-workgroup hand-off per run is never waited for in practice (write-through stores + flag on the
// producer, relaxed poll + device-scope loads on the consumer; a bounded spin sets NVH_DEVERR_HANDOFF and the host
// repeats the batch through the two-kernel path, so a dispatch order that is not the expected one costs time, never
// correctness).
//
// Phases of one frame, wavefront roles (NT = 256: 4 wavefronts; NT = 384: 6 wavefronts):
//   A  side information -> LDS, pair records (staging wavefronts) || Floor1 unwrap + segment list (floor wavefronts)
//   B  residue chain walk (+ inverse coupling) -> barrier -> floor curve multiply          all wavefronts
//   C  IMDCT + window + overlap + clip + PCM stores, one wavefront per channel
// A of the next frame runs while C of this one does: the two only share read-only tables.
//
// Contract (host: nvh_launch.hip run_eligible): <= 2 channels, every residue on the pair path, Floor1, block sizes
// 256..2048, consistent window flags (no overlap reaches into a tail), every overlap source is the frame before.
#include <hip/hip_runtime.h>

#include "imdct_wave.h"
#include "kernels_common.h"
#include "spectrum_dev.h"

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

// 16-byte write-through store / device-scope load (sc1): what leaves here is visible to every other compute unit's
// sc1 loads without a cache write-back fence (MI355X guide, inter-workgroup hand-off R1).
__device__ __forceinline__ void store_wt(float* p, float4 v) {
  const f4v t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ float4 load_dev(const float* p) {
  f4v t;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(p) : "memory");
  return make_float4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ float clipf(float v, int clip, int& clipped) {  // Utils.cs:30-43
  if (clip) {
    if (v > .99999994f) { clipped = 1; return 0.99999994f; }
    if (v < -.99999994f) { clipped = 1; return -0.99999994f; }
  }
  return v;
}

// Emission geometry of one frame (what ReadNextPacket leaves in _prevPacketStart/_prevPacketEnd, StreamDecoder.cs:417-463).
struct Emit {
  int start;      // first emitted block position
  int count;      // emitted samples per channel
  float* out;     // pcm + out_pos * channels
};

// Four consecutive samples of channel c, sample times t0 .. t0+3 of the frame's emission, into the interleaved PCM.
__device__ __forceinline__ void emit4(const Emit& e, int nch, int c, int t0, float4 v, int clip, int& clipped) {
  v.x = clipf(v.x, clip, clipped); v.y = clipf(v.y, clip, clipped);
  v.z = clipf(v.z, clip, clipped); v.w = clipf(v.w, clip, clipped);
  float* o = e.out + (long long)t0 * nch + c;
  if (t0 + 3 < e.count) {
    if (nch == 1 && ((reinterpret_cast<unsigned long long>(o) & 15ull) == 0)) {
      *reinterpret_cast<float4*>(o) = v;
    } else {
      o[0] = v.x; o[nch] = v.y; o[2 * nch] = v.z; o[3 * nch] = v.w;
    }
  } else {  // the end-of-stream trim ends anywhere (StreamDecoder.cs:429-437)
    if (t0 < e.count) o[0] = v.x;
    if (t0 + 1 < e.count) o[nch] = v.y;
    if (t0 + 2 < e.count) o[2 * nch] = v.z;
  }
}

}  // namespace


template <int NT, int NCH>
__device__ __forceinline__ void run_body(const NvhDevSetup& S, const NvhDevBatch& Bt, const NvhRunArgs& R, int* __restrict__ err,
                                         int cap_pass, int cap_ops, int cap_ent, float* smem) {
  constexpr int NW = NT / 64;
  constexpr int nch = NCH;  // compile-time: interleave strides, channel offsets and the stereo-only branches fold
  const int tid0 = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid0 >> 6), lane0 = tid0 & 63;
  const int hmax = S.block1 >> 1;
  // ---- LDS map (4-byte words) ----
  float* s_db = smem;
  FloorScratch* fs = reinterpret_cast<FloorScratch*>(smem + 256);
  uint32_t* s_pass = reinterpret_cast<uint32_t*>(fs + nch);
  NvhDevBook* s_books = reinterpret_cast<NvhDevBook*>(s_pass + cap_pass * 16);
  uint32_t* s_lat = reinterpret_cast<uint32_t*>(reinterpret_cast<float*>(s_books) + S.nbooks * 8);
  NvhResOp* s_ops = reinterpret_cast<NvhResOp*>(s_lat + ((S.lattice_words + 3) & ~3));
  uint4* s_oprec = reinterpret_cast<uint4*>(reinterpret_cast<float*>(s_ops) + cap_ops * 2);
  uint16_t* s_link = reinterpret_cast<uint16_t*>(reinterpret_cast<float*>(s_oprec) + cap_ops * 4);
  uint16_t* s_ent = s_link + cap_ops;
  // spectrum area with its own padding on both sides (the IMDCT slices spill n/16 floats), then the parked head
  float* spec = reinterpret_cast<float*>(s_ent) + (cap_ent >> 1) + (S.block1 >> 4);
  float* park0 = spec + nch * hmax + (S.block1 >> 4);  // two parked heads: the one being completed / emitted and the one being written

  // wavefront roles
  const int floor_w0 = (NW >= 6) ? 2 : 0;                 // first floor wavefront (one per channel)
  const int stage_w0 = floor_w0 + nch;                    // staging wavefronts: [stage_w0, NW)
  const int nstage = NW - stage_w0;

  const int f_lo = (int)blockIdx.x * R.run_len;
  int f_hi = f_lo + R.run_len - 1;
  if (f_hi > Bt.nframes - 1) f_hi = Bt.nframes - 1;
  if (f_lo > f_hi) return;

  // ---- once per workgroup: tables that do not depend on the frame ----
  int tid = tid0, lane = lane0;
  for (int i = tid; i < 256; i += NT) s_db[i] = k_inverse_db[i];
  {
    const uint4* gb = reinterpret_cast<const uint4*>(S.books);
    for (int i = tid; i < S.nbooks * 2; i += NT) reinterpret_cast<uint4*>(s_books)[i] = gb[i];
    for (int i = tid; i < S.lattice_words; i += NT) s_lat[i] = S.lattice[i];
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int i = tid; i < (nch * hmax) >> 2; i += NT) reinterpret_cast<float4*>(spec)[i] = z;  // Mapping.cs:108, first frame
  }
  __syncthreads();  // the pair records of phase A read the codebook directory

  // ---- phase A of frame f: side information -> LDS (staging wavefronts), floors (floor wavefronts) ----
  auto phase_a = [&](int f, int lane) {
    const NvhFrame fr = Bt.frames[f];
    if (fr.n == 0) return;
    const int half = fr.n >> 1;
    const int npass = (int)(fr.pass_end - fr.pass_begin);
    if ((int)fr.op_count > cap_ops || (int)fr.ent_count + 7 > cap_ent || npass > cap_pass) __builtin_trap();
    if (wv >= floor_w0 && wv < floor_w0 + nch) {
      const int c = wv - floor_w0;
      const NvhChan* chans = Bt.chans + (long long)f * nch;
      const FloorLane L = load_floor_lane(S, Bt, chans, c, nch, lane);
      floor_prepare(&fs[c], L, lane, half, err, S.recip);
    } else if (wv >= stage_w0) {
      const int sw = wv - stage_w0;
      const unsigned ent_shift = fr.ent_begin & 7u;
      if (sw == nstage - 1) {
        // op side: ops, links, pass records, then the pair records (wavefront-local ordering only)
        const uint2* go = reinterpret_cast<const uint2*>(Bt.ops + fr.op_begin);
        for (int i = lane; i < (int)fr.op_count; i += 64) reinterpret_cast<uint2*>(s_ops)[i] = go[i];
        const uint16_t* gl = Bt.op_link + fr.op_begin;
        for (int i = lane; i < (int)fr.op_count; i += 64) s_link[i] = gl[i];
        for (int p = lane; p < npass; p += 64) {
          const NvhResPass* gp = Bt.passes + fr.pass_begin + p;
          const NvhDevResidue* Rp = &S.residues[gp->residue];
          uint32_t* P = s_pass + p * 16;
          P[0] = (uint32_t)gp->residue;
          for (int k = 0; k <= NVH_MAX_STAGES; ++k) P[1 + k] = gp->op_begin[k] - fr.op_begin;
          P[10] = (uint32_t)Rp->type | (Rp->pair_path ? 0x100u : 0u);
          P[11] = (uint32_t)Rp->real_channels;
          P[12] = (uint32_t)Rp->partition_size;
          P[13] = Rp->hp_magic;
          P[14] = Rp->rch_magic;
          P[15] = (uint32_t)Rp->begin;
        }
        sp_wave_sync();
        for (int ps = 0; ps < npass; ++ps) {
          const uint32_t* P = s_pass + ps * 16;
          const unsigned rflags = __builtin_amdgcn_readfirstlane(P[10]);
          const int o_end = __builtin_amdgcn_readfirstlane((int)P[1 + NVH_MAX_STAGES]);
          const unsigned rch = __builtin_amdgcn_readfirstlane(P[11]), psz = __builtin_amdgcn_readfirstlane(P[12]);
          const unsigned rch_magic = __builtin_amdgcn_readfirstlane(P[14]), rbegin = __builtin_amdgcn_readfirstlane(P[15]);
          for (int o = __builtin_amdgcn_readfirstlane((int)P[1]) + lane; o < o_end; o += 64) {
            const NvhResOp op = s_ops[o];
            const NvhDevBook bk = s_books[op.book];
            const unsigned offset = rbegin + (unsigned)op.partition * psz;
            const unsigned xbase = ((rflags & 0xFFu) == 2 && rch > 1) ? __umulhi(offset, rch_magic) : offset;
            uint4 rec;
            rec.x = (op.ent_off - fr.ent_begin) | (xbase << 16);
            rec.y = bk.lat_off | (bk.lat_values << 16);
            rec.z = bk.lat_magic;
            rec.w = bk.dim | ((unsigned)op.channel << 8) | (bk.dim_magic16 << 16);
            s_oprec[o] = rec;
          }
        }
      }
      if (nstage == 1 || sw < nstage - 1) {
        // entry slice, from its enclosing 16-byte boundary
        const int st = (nstage == 1 ? 0 : sw) * 64 + lane, sn = (nstage == 1 ? 1 : nstage - 1) * 64;
        const uint4* ge = reinterpret_cast<const uint4*>(Bt.entries + (fr.ent_begin - ent_shift));
        const int nvec = (int)((ent_shift + fr.ent_count + 7u) >> 3);
        for (int i = st; i < nvec; i += sn) reinterpret_cast<uint4*>(s_ent)[i] = ge[i];
      }
    }
  };

#ifdef NVH_DEBUG
  // [workgroup][wavefront][frame slot 0..7][8 stamps]: 0 loop top (after barrier), 1 sweep done, 2 tail done, 3 C / A done
#define RUN_T(slot, k)                                                                                                         \
  do {                                                                                                                         \
    if (R.dbg && lane0 == 0 && (slot) < 8) R.dbg[(((long long)blockIdx.x * NW + wv) * 8 + (slot)) * 8 + (k)] = wall_clock64(); \
  } while (0)
#else
#define RUN_T(slot, k) do { } while (0)
#endif
  RUN_T(0, 7);
  phase_a(f_hi, lane0);
  RUN_T(0, 6);

  int clipped = 0;
  // Two frames are parked at any time (one n/2-float head per channel each, two LDS buffers):
  //   wt  computed in the previous iteration, its head still WAITS for the tail of the frame computed now;
  //   dn  complete (DONE), goes out as PCM after the next workgroup barrier, by all wavefronts.
  struct Parked {
    bool on;
    int buf;                 // which park buffer
    Emit e;
    int half, start, ov_src, ov_len, ov_frame;
  };
  Parked wt{false, 0, Emit{0, 0, nullptr}, 0, 0, 0, 0, -1}, dn = wt;
  int prev_half = hmax;  // layout the spectrum area was last cleared for

  // PCM of a parked frame: block positions [emit.start, min(emit end, half)) from its parked head (positions beyond the
  // half were emitted when the block was computed: nothing ever overlaps them).  All wavefronts; a lane takes four sample
  // times of every channel and writes nch 16-byte vectors that are contiguous in the interleaved PCM (one scattered 4-byte
  // store per sample and channel costs 16x the L2 write requests: measured 125 us per batch instead of the two-kernel 54).
  auto emit_frame = [&](const Parked& p, int tid) {
    const int lo_end = p.e.start + p.e.count < p.half ? p.e.start + p.e.count : p.half;
    const int cnt = lo_end - p.e.start;  // samples per channel that come from the parked head
    if (cnt <= 0) return;
    const float* h = park0 + p.buf * (nch * hmax) + p.e.start;
    const int groups = cnt >> 2;
    const bool aligned = ((reinterpret_cast<unsigned long long>(p.e.out) & 15ull) == 0);
    for (int g = tid; g < groups; g += NT) {
      float flat[4 * nch];
#pragma unroll
      for (int c = 0; c < nch; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(h + c * p.half + 4 * g);
        flat[0 * nch + c] = clipf(v.x, R.clip, clipped);
        flat[1 * nch + c] = clipf(v.y, R.clip, clipped);
        flat[2 * nch + c] = clipf(v.z, R.clip, clipped);
        flat[3 * nch + c] = clipf(v.w, R.clip, clipped);
      }
      float* o = p.e.out + 4 * g * nch;
      if (aligned) {
#pragma unroll
        for (int k = 0; k < nch; ++k) reinterpret_cast<float4*>(o)[k] = make_float4(flat[4 * k], flat[4 * k + 1], flat[4 * k + 2], flat[4 * k + 3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4 * nch; ++k) o[k] = flat[k];
      }
    }
    const int rest = cnt & 3;  // the last, partial group of an end-of-stream trim
    if (rest && tid < rest * nch) {
      const int t = (groups << 2) + tid / nch, c = tid % nch;
      p.e.out[t * nch + c] = clipf(h[c * p.half + t], R.clip, clipped);
    }
  };

  for (int f = f_hi; f >= f_lo; --f) {
    // Per-lane index arithmetic of the phases below (LDS addresses of the IMDCT passes, table offsets, ...) depends on the
    // lane only; left alone the optimiser hoists all of it out of the frame loop and keeps it live -- ~200 spilled
    // registers.  An opaque copy of the lane id per iteration keeps every such value inside its phase.
    tid = tid0;
    lane = lane0;
    asm volatile("" : "+v"(tid), "+v"(lane));
    const NvhFrame fr = Bt.frames[f];
    const int n = fr.n, half = n >> 1;
    if (n == 0) {
      // pseudo-frame (only ever the batch's first): the carried block's tail goes out as it is (StreamDecoder.cs:352-356)
      __syncthreads();  // phase C of the frame above is complete; after a drain nothing overlaps it (its ov_len is 0)
      if (dn.on) emit_frame(dn, tid);
      if (wt.on) emit_frame(wt, tid);
      dn.on = wt.on = false;
      if (wv < nch) {
        const int c = wv;
        Emit e{0, fr.emit_count, R.pcm + fr.out_pos * nch};
        const float* src = R.carry + (long long)c * S.block1 + fr.ov_src;
        for (int t = 4 * lane; t < fr.emit_count; t += 256) {
          float4 v;
          v.x = src[t];
          v.y = t + 1 < fr.emit_count ? src[t + 1] : 0.0f;
          v.z = t + 2 < fr.emit_count ? src[t + 2] : 0.0f;
          v.w = t + 3 < fr.emit_count ? src[t + 3] : 0.0f;
          emit4(e, nch, c, t, v, R.clip, clipped);
        }
      }
      continue;
    }
    const NvhDevMapping mp = S.mappings[fr.mapping];
    const int npass = (int)(fr.pass_end - fr.pass_begin);
    const unsigned ent_shift = fr.ent_begin & 7u;
    const uint16_t* ent = s_ent + ent_shift;
    __syncthreads();  // phase A of this frame (and phase C of the one above it) are complete
    RUN_T(f_hi - f, 0);
    if (dn.on) {  // completed by phase C of the previous iteration; its buffer is free again before this iteration's phase C
      emit_frame(dn, tid);
      dn.on = false;
    }
    if (half != prev_half) {
      // block size changed: the channel regions of the spectrum area moved, the per-channel clears below do not fit
      const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      for (int i = tid; i < (nch * hmax) >> 2; i += NT) reinterpret_cast<float4*>(spec)[i] = z;
      __syncthreads();
    }
    prev_half = half;

    // ---- phase B: residue chain walk (kernels_spectrum.hip, FAST sweep) ----
    int cpl_mg = 0;
    bool couple = false;
    if (nch == 2 && mp.coupling_steps == 1) {
      cpl_mg = S.coupling[mp.coupling_off];
      couple = (fr.exec_mask & 3u) != 0;
    }
    const bool sweep_couples = couple && npass == 1 && (__builtin_amdgcn_readfirstlane(s_pass[10]) & 0xFFu) == 2 &&
                               __builtin_amdgcn_readfirstlane(s_pass[11]) == 2;
    for (int ps = 0; ps < npass; ++ps) {
      const uint32_t* P = s_pass + ps * 16;
      const unsigned rflags = __builtin_amdgcn_readfirstlane(P[10]);
      const int rtype = (int)(rflags & 0xFFu);
      const int psize = __builtin_amdgcn_readfirstlane((int)P[12]);
      const unsigned rch = __builtin_amdgcn_readfirstlane(P[11]);
      const unsigned hp_magic = __builtin_amdgcn_readfirstlane(P[13]), rch_magic = __builtin_amdgcn_readfirstlane(P[14]);
      const unsigned hp = (unsigned)psize >> 1;
      const unsigned o0 = __builtin_amdgcn_readfirstlane(P[1]), o1 = __builtin_amdgcn_readfirstlane(P[1 + NVH_MAX_STAGES]);
      const unsigned total = (o1 - o0) * hp;
      for (unsigned idx = tid; idx < total; idx += NT) {
        const unsigned oq = hp > 1 ? __umulhi(idx, hp_magic) : idx;
        const unsigned i2 = idx - oq * hp, i = i2 << 1;
        unsigned o = o0 + oq;
        unsigned link = s_link[o];
        if (link & 0x8000u) continue;  // not a chain head
        uint4 rec = s_oprec[o];
        const unsigned xbase = rec.x >> 16;
        unsigned c0, x0, c1, x1;
        if (rtype == 1 || rch == 1) {
          c0 = c1 = (rec.w >> 8) & 0xFFu;
          x0 = xbase + i;
          x1 = x0 + 1;
        } else {  // rch == 2 (the contract admits at most two channels)
          c0 = 0; c1 = 1;
          x0 = x1 = xbase + i2;
        }
        (void)rch_magic;
        const bool in0 = x0 < (unsigned)half, in1 = x1 < (unsigned)half;
        float* p0 = spec + c0 * (unsigned)half + x0;
        float* p1 = spec + c1 * (unsigned)half + x1;
        float a0 = in0 ? *p0 : 0.0f, a1 = in1 ? *p1 : 0.0f;
        for (;;) {
          const unsigned dims = rec.w & 0xFFu, lv = rec.y >> 16;
          const unsigned j = (i * (rec.w >> 16)) >> 16;
          const unsigned comp = i - j * dims;
          unsigned q = ent[(rec.x & 0xFFFFu) + j];
          if (q != NVH_ENTRY_SKIP) {
            const uint32_t* lat = s_lat + (rec.y & 0xFFFFu);
            if (comp) q = __umulhi(q, lat[lv + comp]);
            const unsigned q1 = __umulhi(q, rec.z);
            const unsigned d0 = q - q1 * lv;
            const unsigned d1 = q1 - __umulhi(q1, rec.z) * lv;
            a0 = a0 + __uint_as_float(lat[d0]);
            a1 = a1 + __uint_as_float(lat[d1]);
          }
          link &= 0x7FFFu;
          if (link == NVH_LINK_NONE) break;
          o = link;
          rec = s_oprec[o];
          link = s_link[o];
        }
        if (sweep_couples) {
          if (cpl_mg == 0) couple1(a0, a1); else couple1(a1, a0);
        }
        if (in0) *p0 = a0;
        if (in1) *p1 = a1;
      }
      __syncthreads();
    }
    if (npass == 0) __syncthreads();
    RUN_T(f_hi - f, 1);

    // ---- phase B, second half: inverse coupling where the sweep could not do it, floor curve multiply ----
    {
      const bool tail_couples = couple && !sweep_couples;
      const int md0 = __builtin_amdgcn_readfirstlane(fs[0].mode), md1 = nch == 2 ? __builtin_amdgcn_readfirstlane(fs[1].mode) : 0;
      if (nch == 2 && !tail_couples) {
        constexpr int TS = 8;
        if (tid < 256) {
          const int c = tid >> 7;
          const int md = c ? md1 : md0;
          float* sp = spec + c * half;
          if (md != 0) {
            for (int x0 = (tid & 127) * TS; x0 < half; x0 += 128 * TS) {
              float r[TS], m[TS];
              if (md == 1) {
#pragma unroll
                for (int q = 0; q < TS; q += 4) *reinterpret_cast<float4*>(r + q) = *reinterpret_cast<const float4*>(sp + x0 + q);
                floor_walk<TS>(&fs[c], s_db, x0, m);
#pragma unroll
                for (int q = 0; q < TS; ++q) r[q] = r[q] * m[q];
              } else {
#pragma unroll
                for (int q = 0; q < TS; ++q) r[q] = 0.0f;  // Floor1.cs:218-221
              }
#pragma unroll
              for (int q = 0; q < TS; q += 4) *reinterpret_cast<float4*>(sp + x0 + q) = *reinterpret_cast<float4*>(r + q);
            }
          }
        }
      } else {
        constexpr int TB = 4;
        for (int x0 = tid * TB; x0 < half; x0 += NT * TB) {
          float r0[TB], r1[TB], m[TB];
          *reinterpret_cast<float4*>(r0) = *reinterpret_cast<const float4*>(spec + x0);
          if (nch == 2) *reinterpret_cast<float4*>(r1) = *reinterpret_cast<const float4*>(spec + half + x0);
          if (tail_couples) {
#pragma unroll
            for (int q = 0; q < TB; ++q) {
              if (cpl_mg == 0) couple1(r0[q], r1[q]); else couple1(r1[q], r0[q]);
            }
          }
          if (md0 == 1) {
            floor_walk<TB>(&fs[0], s_db, x0, m);
#pragma unroll
            for (int q = 0; q < TB; ++q) r0[q] = r0[q] * m[q];
          } else if (md0 == 2) {
#pragma unroll
            for (int q = 0; q < TB; ++q) r0[q] = 0.0f;
          }
          *reinterpret_cast<float4*>(spec + x0) = *reinterpret_cast<float4*>(r0);
          if (nch == 2) {
            if (md1 == 1) {
              floor_walk<TB>(&fs[1], s_db, x0, m);
#pragma unroll
              for (int q = 0; q < TB; ++q) r1[q] = r1[q] * m[q];
            } else if (md1 == 2) {
#pragma unroll
              for (int q = 0; q < TB; ++q) r1[q] = 0.0f;
            }
            *reinterpret_cast<float4*>(spec + half + x0) = *reinterpret_cast<float4*>(r1);
          }
        }
      }
    }
    __syncthreads();
    RUN_T(f_hi - f, 2);

    // ---- phase C (one wavefront per channel) || phase A of the frame below (the others) ----
    const bool next_a = f > f_lo;
    if (wv < nch) {
      const int c = wv;
      float* Sc = spec + c * half;                       // the channel's spectrum, then its windowed head, natural order
      const bool exec = ((fr.exec_mask >> c) & 1u) != 0;
      if (exec) {
        // IMDCT in place; its two independent output quarters y[0, n/4) and y[n/2, 3n/4) stay in the channel's own LDS
        // region (Sc[0, n/4) and Sc[n/4, n/2)); the other two follow from y[n/2-1-x] = -y[x], y[n-1-x] = y[n/2+x]
        // (Mdct.cs:275-303)
        const int n4 = n >> 2;
        auto keep = [=](int slot, int idx, float4 v) {
          if ((slot & 1) == 0) *reinterpret_cast<float4*>(Sc + (idx < half ? idx : idx - n4)) = v;
        };
        float* scratch = Sc - (nch - 1 - c) * (n >> 4);
        const int sl = fr.mdct_slot;
        const float* A = S.mdct_a[sl];
        const float* B = S.mdct_b[sl];
        const float* C = S.mdct_c[sl];
        const float* TW = S.mdct_tw[sl];
        switch (n) {
          case 256: imdct_wave_sink<8, false, decltype(keep), true>(Sc, nullptr, scratch, A, B, C, TW, lane, keep); break;
          case 512: imdct_wave_sink<9, false, decltype(keep), true>(Sc, nullptr, scratch, A, B, C, TW, lane, keep); break;
          case 1024: imdct_wave_sink<10, false, decltype(keep), true>(Sc, nullptr, scratch, A, B, C, TW, lane, keep); break;
          case 2048: imdct_wave_sink<11, false, decltype(keep), true>(Sc, nullptr, scratch, A, B, C, TW, lane, keep); break;
          default: __builtin_trap();
        }
        sp_wave_sync();
      }
      // (everything the loops below need is derived here, after the transform: nothing of it is live across it)
      const float* __restrict__ w = S.windows + fr.window_off;
      const Emit me{fr.emit_start, fr.emit_count, R.pcm + fr.out_pos * nch};
      const int emit_end = fr.emit_start + fr.emit_count;
      const bool publish = f == f_hi && f + 1 < Bt.nframes;   // the run above needs this block's tail
      const bool to_carry = f == R.last_decoded && R.carry_out != nullptr;
      float* tail_plane = R.tails + ((long long)f * nch + c) * S.block1;
      float* carry_plane = R.carry_out + (long long)c * S.block1;
      float* wt_head = park0 + wt.buf * (nch * hmax) + c * wt.half;
      const bool feeds_parked = wt.on && wt.ov_len > 0 && wt.ov_frame == f;
      const int my_buf = wt.on ? 1 - wt.buf : 0;  // the buffer of the frame that went out at the top of this iteration
      // windowed block values at positions d .. d+3 (Mode.cs:160-166); a channel that does not execute is
      // [residue | zeros] (Mapping.cs:192-196, quirk B-4)
      auto block4 = [&](int d) -> float4 {
        float4 y;
        if (exec) {
          const int n4 = n >> 2;
          if (d < n4) {
            y = *reinterpret_cast<const float4*>(Sc + d);
          } else if (d < half) {
            const float4 r = *reinterpret_cast<const float4*>(Sc + (half - 4 - d));
            y = make_float4(-r.w, -r.z, -r.y, -r.x);
          } else if (d < half + n4) {
            y = *reinterpret_cast<const float4*>(Sc + (d - n4));
          } else {
            const float4 r = *reinterpret_cast<const float4*>(Sc + (n4 + n - 4 - d));
            y = make_float4(r.w, r.z, r.y, r.x);
          }
        } else {
          y = d < half ? *reinterpret_cast<const float4*>(Sc + d) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 ww = *reinterpret_cast<const float4*>(w + d);
        return make_float4(y.x * ww.x, y.y * ww.y, y.z * ww.z, y.w * ww.w);
      };
      // second half of the block: completes the parked head of the frame above, goes out as it is where this frame
      // drains its own tail, is published for the run above, becomes the carried block
#pragma unroll 2
      for (int d = half + 4 * lane; d < n; d += 256) {
        const float4 v = block4(d);
        if (to_carry) *reinterpret_cast<float4*>(carry_plane + d) = v;
        if (d >= me.start && d < emit_end) emit4(me, nch, c, d - me.start, v, R.clip, clipped);  // nothing overlaps these
        if (publish) store_wt(tail_plane + d, v);
        if (feeds_parked) {
          const int j = d - wt.ov_src;
          if (j >= 0 && j < wt.ov_len) {  // OverlapBuffers: next[start + j] += previous[prevEnd + j]
            float4* h = reinterpret_cast<float4*>(wt_head + wt.start + j);
            float4 t = *h;
            t.x = t.x + v.x; t.y = t.y + v.y; t.z = t.z + v.z; t.w = t.w + v.w;
            *h = t;
          }
        }
      }
      if (publish) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every write-through store of this wavefront has left
        if (lane == 0) __hip_atomic_store(R.flags + (long long)f * nch + c, R.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // first half: parked until the frame below has been computed
      {
        float* ph = park0 + my_buf * (nch * hmax) + c * half;
#pragma unroll 2
        for (int d = 4 * lane; d < half; d += 256) {
          const float4 v = block4(d);
          if (to_carry) *reinterpret_cast<float4*>(carry_plane + d) = v;
          *reinterpret_cast<float4*>(ph + d) = v;
        }
        sp_wave_sync();
        // leave the channel's spectrum region cleared for the next frame (Mapping.cs:108)
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int i = 4 * lane; i < half; i += 256) *reinterpret_cast<float4*>(Sc + i) = z;
      }
      if (NW < 6 && next_a) phase_a(f - 1, lane);  // four-wavefront form: the channel wavefronts unwrap the next floors themselves
    } else if (next_a) {
      phase_a(f - 1, lane);
    }
    RUN_T(f_hi - f, 3);
    {
      const int my_buf = wt.on ? 1 - wt.buf : 0;
      dn = wt;  // complete once this iteration's phase C is (the barrier at the top of the next iteration)
      wt = Parked{true, my_buf, Emit{fr.emit_start, fr.emit_count, R.pcm + fr.out_pos * nch}, half, fr.start, fr.ov_src, fr.ov_len, fr.ov_frame};
    }
  }

  // ---- run end: the parked head of the run's first frame gets the tail of the frame before it ----
  tid = tid0;
  lane = lane0;
  asm volatile("" : "+v"(tid), "+v"(lane));
  __syncthreads();  // the last phase C is complete
  if (dn.on) emit_frame(dn, tid);
  if (wt.on && wv < nch && wt.ov_len > 0 && (wt.ov_frame == -2 || wt.ov_frame >= 0)) {
    const int c = wv;
    float* h = park0 + wt.buf * (nch * hmax) + c * wt.half + wt.start;
    if (wt.ov_frame == -2) {
      const float* src = R.carry + (long long)c * S.block1 + wt.ov_src;
      for (int j = 4 * lane; j < wt.ov_len; j += 256) {
        float4 t = *reinterpret_cast<float4*>(h + j);
        const float4 v = *reinterpret_cast<const float4*>(src + j);
        t.x = t.x + v.x; t.y = t.y + v.y; t.z = t.z + v.z; t.w = t.w + v.w;
        *reinterpret_cast<float4*>(h + j) = t;
      }
    } else {
      // published by the neighbouring workgroup when it STARTED; poll anyway (bounded), then device-scope loads
      const unsigned* flag = R.flags + (long long)wt.ov_frame * nch + c;
      bool ok = true;
      if (lane == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != R.epoch) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > 4000000u) { ok = false; break; }
        }
        if (!ok) atomicOr(err, NVH_DEVERR_HANDOFF);
      }
      ok = __shfl((int)ok, 0) != 0;
      if (ok) {
        const float* src = R.tails + ((long long)wt.ov_frame * nch + c) * S.block1 + wt.ov_src;
        for (int j = 4 * lane; j < wt.ov_len; j += 256) {
          float4 t = *reinterpret_cast<float4*>(h + j);
          const float4 v = load_dev(src + j);
          t.x = t.x + v.x; t.y = t.y + v.y; t.z = t.z + v.z; t.w = t.w + v.w;
          *reinterpret_cast<float4*>(h + j) = t;
        }
      }
    }
  }
  __syncthreads();
  if (wt.on) emit_frame(wt, tid);
  RUN_T(7, 5);
  report_clipped(clipped, R.clipped_flag);
}

// The setup and batch parameter blocks are read from device memory, not passed by value: their ~40 pointers would
// otherwise sit in scalar registers for the whole frame loop (116 scalar spills measured that way).
// Register budget: 6 wavefronts per SIMD (80 VGPRs): four 6-wavefront workgroups, or six 4-wavefront ones, per CU.
#define NVH_RUN_KERNEL(name, NT, NCH)                                                                                          \
  extern "C" __global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(6, 6)))                                   \
  name(const NvhDevSetup* Sg, const NvhDevBatch* Bg, NvhRunArgs R, int* __restrict__ err, int cap_pass, int cap_ops, int cap_ent) { \
    extern __shared__ __attribute__((aligned(16))) float smem[];                                                               \
    run_body<NT, NCH>(*Sg, *Bg, R, err, cap_pass, cap_ops, cap_ent, smem);                                                     \
  }
NVH_RUN_KERNEL(k_run4_c1, 256, 1)
NVH_RUN_KERNEL(k_run4_c2, 256, 2)
NVH_RUN_KERNEL(k_run6_c1, 384, 1)
NVH_RUN_KERNEL(k_run6_c2, 384, 2)
