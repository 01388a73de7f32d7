// kernels_spectrum2.hip -- k_spectrum_imdct with a frame loop: one workgroup decodes several frames, and while two of its
// wavefronts run the inverse MDCT of frame k the other two stage the side information and unwrap the floors of frame k+1.
//
//   Array.Clear + IResidue.Decode adds      Mapping.cs:108,133; Residue1.cs:8-26, Residue2.cs:23-47
//   inverse square-polar coupling            Mapping.cs:137-182
//   IFloor.Apply (Floor1)                    Floor1.cs:186-341
//   IMdct.Reverse                            Mdct.cs:65-313 (imdct_wave.h)
//
// Same contract, LDS map (+ one n/16-float pad in front of the spectrum), register budget (64 VGPRs: 8 workgroups per CU)
// and output (the compact IMDCT quarters k_ola_compact reads) as k_spectrum_imdct in kernels_spectrum.hip.  Measured there
// (profiles/r02_run_phases.txt): a workgroup's life is the global-load chain of the frame record (~4k cycles), the staging
// / floor phase (~7k), the chain walk (~8k), the floor multiply (~7.5k) and the inverse MDCT (~7k) on two of four
// wavefronts -- a third of it with half the workgroup idle or waiting on memory.  Here the staging of the next frame fills
// that third:
//
//   wavefront   0, 1                    2                         3
//   prologue    floors (0)              entries                   ops, pair records         of the first frame
//   per frame   ------------- chain walk, barrier, floor multiply, barrier -------------
//               IMDCT ch 0 / ch 1       ops, pair records         entries, floors           of the NEXT frame
//
// The floors of both channels are unwrapped by ONE wavefront side by side (floor_prepare<DUAL>, 32 lanes each) when
// neither has more than 32 posts, one after the other otherwise.  Mono: wavefront 1 takes the floor.
#include <hip/hip_runtime.h>

#include "imdct_wave.h"
#include "kernels_common.h"
#include "spectrum_dev.h"

template <int NCH>
__device__ __forceinline__ void frames_body(const NvhDevSetup& S, const NvhDevBatch& Bt, float* __restrict__ work, int* __restrict__ err,
                                            int cap_pass, int cap_ops, int cap_ent, float* smem NVH_DBG_PARAMS) {
  constexpr int nch = NCH;
  const int tid0 = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid0 >> 6), lane0 = tid0 & 63;
  const int hmax = S.block1 >> 1;
  // ---- LDS map (4-byte words): k_spectrum_imdct's, with the IMDCT padding of channel 0 in front of the spectrum instead
  // of on top of the staging area (which now belongs to the next frame while the transform runs) ----
  float* s_db = smem;
  FloorScratch* fs = reinterpret_cast<FloorScratch*>(smem + 256);
  uint32_t* s_pass = reinterpret_cast<uint32_t*>(fs + nch);
  NvhDevBook* s_books = reinterpret_cast<NvhDevBook*>(s_pass + cap_pass * 16);
  uint32_t* s_lat = reinterpret_cast<uint32_t*>(reinterpret_cast<float*>(s_books) + S.nbooks * 8);
  NvhResOp* s_ops = reinterpret_cast<NvhResOp*>(s_lat + ((S.lattice_words + 3) & ~3));
  uint4* s_oprec = reinterpret_cast<uint4*>(reinterpret_cast<float*>(s_ops) + cap_ops * 2);
  uint16_t* s_link = reinterpret_cast<uint16_t*>(reinterpret_cast<float*>(s_oprec) + cap_ops * 4);
  uint16_t* s_ent = s_link + cap_ops;
  float* spec = reinterpret_cast<float*>(s_ent) + (cap_ent >> 1) + (S.block1 >> 4);

  const int stride = (int)gridDim.x;
  int f = (int)blockIdx.x;
  if (f >= Bt.nframes) return;

  // ---- once per workgroup: tables that do not depend on the frame ----
  int tid = tid0, lane = lane0;
  for (int i = tid; i < 256; i += 256) s_db[i] = k_inverse_db[i];
  {
    const uint4* gb = reinterpret_cast<const uint4*>(S.books);
    for (int i = tid; i < S.nbooks * 2; i += 256) reinterpret_cast<uint4*>(s_books)[i] = gb[i];
    for (int i = tid; i < S.lattice_words; i += 256) s_lat[i] = S.lattice[i];
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int i = tid; i < (nch * hmax) >> 2; i += 256) reinterpret_cast<float4*>(spec)[i] = z;  // Mapping.cs:108, first frame
  }
  __syncthreads();  // the pair records read the codebook directory

  // ---- phase A of frame fa: side information -> LDS, floors.  floor_w / ent_w / ops_w: which wavefront does what ----
  auto phase_a = [&](int fa, int lane, int floor_w, int ent_w, int ops_w) {
    const NvhFrame fr = Bt.frames[fa];
    if (fr.n == 0) return;
    const int half = fr.n >> 1;
    const int npass = (int)(fr.pass_end - fr.pass_begin);
    if ((int)fr.op_count > cap_ops || (int)fr.ent_count + 7 > cap_ent || npass > cap_pass) __builtin_trap();
    if (wv == ent_w) {
      // entry slice, from its enclosing 16-byte boundary
      const unsigned ent_shift = fr.ent_begin & 7u;
      const uint4* ge = reinterpret_cast<const uint4*>(Bt.entries + (fr.ent_begin - ent_shift));
      const int nvec = (int)((ent_shift + fr.ent_count + 7u) >> 3);
      for (int i = lane; i < nvec; i += 64) reinterpret_cast<uint4*>(s_ent)[i] = ge[i];
    }
    if (wv == floor_w) {
      const NvhChan* chans = Bt.chans + (long long)fa * nch;
      if (nch == 2) {
        const FloorLane L2 = load_floor_lane(S, Bt, chans, lane >> 5, nch, lane & 31);
        if (__ballot(L2.pc > 32) == 0ull) {
          floor_prepare<true>(fs, L2, lane, half, err, S.recip);
        } else {
          for (int c = 0; c < nch; ++c) {
            const FloorLane L = load_floor_lane(S, Bt, chans, c, nch, lane);
            floor_prepare(&fs[c], L, lane, half, err, S.recip);
          }
        }
      } else {
        const FloorLane L = load_floor_lane(S, Bt, chans, 0, nch, lane);
        floor_prepare(&fs[0], L, lane, half, err, S.recip);
      }
    }
    if (wv == ops_w) {
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
  };

#ifdef NVH_DEBUG
  // [workgroup][wavefront][frame slot 0..7][8 stamps]: 7 start, 6 first phase A done; per slot 0 loop top (after the
  // barrier), 1 walk done, 2 floor multiply done, 3 transform / next phase A done; slot 7 stamp 5: end
  int dslot = 0;
#define MF_T(slot, k)                                                                                                       \
  do {                                                                                                                      \
    if (dbg && lane0 == 0 && (slot) < 8) dbg[(((long long)blockIdx.x * 4 + wv) * 8 + (slot)) * 8 + (k)] = wall_clock64();  \
  } while (0)
#else
#define MF_T(slot, k) do { } while (0)
#endif
  MF_T(0, 7);
  phase_a(f, lane0, 0, 2, 3);
  MF_T(0, 6);
  int prev_half = hmax;  // layout the spectrum area was last cleared for

  for (; f < Bt.nframes; f += stride) {
    // Per-lane index arithmetic of the phases below (LDS addresses of the IMDCT passes, table offsets, ...) depends on the
    // lane only; left alone the optimiser hoists all of it out of the frame loop and keeps it live across every phase.
    // An opaque copy of the lane id per iteration keeps every such value inside its phase.
    tid = tid0;
    lane = lane0;
    asm volatile("" : "+v"(tid), "+v"(lane));
    const NvhFrame fr = Bt.frames[f];
    const int n = fr.n, half = n >> 1;
    const int fnext = f + stride;
    const bool has_next = fnext < Bt.nframes;
    if (n == 0) {  // drain pseudo-frame: nothing to compute (k_ola_compact emits the carried tail)
      if (has_next) phase_a(fnext, lane, 0, 2, 3);
      continue;
    }
    const NvhDevMapping mp = S.mappings[fr.mapping];
    const int npass = (int)(fr.pass_end - fr.pass_begin);
    const unsigned ent_shift = fr.ent_begin & 7u;
    const uint16_t* ent = s_ent + ent_shift;
    __syncthreads();  // phase A of this frame and the transform of the one before it are complete
    MF_T(dslot, 0);
    if (half != prev_half) {
      // block size changed: the channel regions of the spectrum area moved, the per-channel clears below do not fit
      const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      for (int i = tid; i < (nch * hmax) >> 2; i += 256) reinterpret_cast<float4*>(spec)[i] = z;
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
      for (unsigned idx = tid; idx < total; idx += 256) {
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

    MF_T(dslot, 1);
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
        for (int x0 = tid * TB; x0 < half; x0 += 256 * TB) {
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

    MF_T(dslot, 2);
    // ---- inverse MDCT (one wavefront per channel) || phase A of the workgroup's next frame (the others) ----
    if (wv < nch) {
      const int c = wv;
      float* Sc = spec + c * half;
      float* out = work + ((long long)f * nch + c) * S.block1;
      if (!((fr.exec_mask >> c) & 1u)) {
        // Mapping.cs:192-196: the residue stays in [0, n/2) (k_ola_compact windows it); its tail quarter is zero
        for (int i = lane * 4; i < half; i += 256) *reinterpret_cast<float4*>(out + i) = *reinterpret_cast<const float4*>(Sc + i);
        for (int i = lane * 4; i < (half >> 1); i += 256) *reinterpret_cast<float4*>(out + half + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      } else {
        float* scratch = Sc - (nch - 1 - c) * (n >> 4);
        const int sl = fr.mdct_slot;
        const float* A = S.mdct_a[sl];
        const float* B = S.mdct_b[sl];
        const float* C = S.mdct_c[sl];
        const float* TW = S.mdct_tw[sl];
        switch (n) {
          case 256: imdct_wave<8, false, true, true>(Sc, out, nullptr, scratch, A, B, C, TW, lane); break;
          case 512: imdct_wave<9, false, true, true>(Sc, out, nullptr, scratch, A, B, C, TW, lane); break;
          case 1024: imdct_wave<10, false, true, true>(Sc, out, nullptr, scratch, A, B, C, TW, lane); break;
          case 2048: imdct_wave<11, false, true, true>(Sc, out, nullptr, scratch, A, B, C, TW, lane); break;
          default: __builtin_trap();
        }
      }
      if (has_next) {
        // leave the channel's spectrum region cleared for the next frame (Mapping.cs:108)
        sp_wave_sync();
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int i = 4 * lane; i < half; i += 256) *reinterpret_cast<float4*>(Sc + i) = z;
      }
    } else if (has_next) {
      if (nch == 2) phase_a(fnext, lane, 3, 3, 2);
      else phase_a(fnext, lane, 1, 3, 2);
    }
    MF_T(dslot, 3);
#ifdef NVH_DEBUG
    ++dslot;
#endif
  }
  MF_T(7, 5);
}

// The setup and batch parameter blocks are read from device memory, not passed by value: their ~40 pointers would
// otherwise sit in scalar registers for the whole frame loop.
#define NVH_FRAMES_KERNEL(name, NCH)                                                                                           \
  extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))                                  \
  name(const NvhDevSetup* Sg, const NvhDevBatch* Bg, float* __restrict__ work, int* __restrict__ err, int cap_pass, int cap_ops, \
       int cap_ent NVH_DBG_PARAMS) {                                                                                           \
    extern __shared__ __attribute__((aligned(16))) float smem[];                                                               \
    frames_body<NCH>(*Sg, *Bg, work, err, cap_pass, cap_ops, cap_ent, smem NVH_DBG_ARGS);                                                   \
  }
NVH_FRAMES_KERNEL(k_spectrum_imdct2_c1, 1)
NVH_FRAMES_KERNEL(k_spectrum_imdct2_c2, 2)
