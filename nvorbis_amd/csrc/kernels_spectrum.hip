// kernels_spectrum.hip -- fused spectrum kernel: residue vector adds + inverse coupling + floor apply with the
// whole frame's spectrum resident in LDS (one workgroup per frame, all channels).
//
//   Array.Clear + IResidue.Decode adds   Mapping.cs:108,133; Residue0.cs:180-201, Residue1.cs:8-26, Residue2.cs:23-47
//   inverse square-polar coupling         Mapping.cs:137-182
//   IFloor.Apply                          Floor1.cs:186-341 (UnwrapPosts :224-297), Floor0.cs:152-212
//
//   IMdct.Reverse (k_spectrum_imdct)       Mdct.cs:65-313, via imdct_wave.h
//
// Kernels: k_spectrum_imdct / k_spectrum (pair-path + fused-tail stream shapes: <= 2 channels, Floor1, lattice books;
// with / without the inverse MDCT behind it), k_spectrum_gen (any other Floor1 shape), k_spectrum_f0 (Floor0).
// Output: work[frame][ch] = the vector IMdct.Reverse consumes in [0, n/2) (or, for a channel that does not execute,
// the raw residue -- quirk B-4); k_spectrum_imdct writes the compact IMDCT output k_ola_compact expects instead.
//
// The kernel moves ~2 KB of side information per frame and is bound by instruction issue and LDS / L2 latency, not
// by HBM (DESIGN.md section 6 has the counters).  Its structure aims at few instructions, a small LDS footprint
// (8 workgroups per CU) and short dependent chains:
//   * side information (ops, op links, entries, codebook directory, lattice pool, pass records) is staged into LDS
//     with 16-byte copies by two wavefronts while the other two unwrap the floor posts; all index divisions are
//     exact reciprocal multiplies prepared by the host;
//   * lattice codebooks (every book libvorbis writes) never touch their VQ table: a lane peels two base-
//     lat_values digits off the entry number and adds two bins ("pair path"); in the fast kernels it walks the
//     host-linked chain of ops of one partition through all cascade stages with the sums in registers;
//   * for mono / stereo Floor1 streams coupling, floor render and the store are one pass: a lane owns 4 consecutive
//     bins of every channel in registers ("fused tail");
//   * k_spectrum_imdct then runs the wavefront IMDCT in place on the LDS spectrum, one wavefront per channel.
// Bit-exactness: residue adds happen in the reference's stage order per element (general kernels: one barrier per
// stage; chain walk: program order inside the owning lane); all float expressions are single operations;
// -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "imdct_wave.h"
#include "kernels_common.h"
#include "spectrum_dev.h"


// LDS map (dynamic, 4-byte words):
//   [ s_db 256 | (FLOOR0: s_coeff 256) | FloorScratch x min(channels, SP_GROUP) | pass records cap_pass*16 |
//     books nbooks*8 | lattice pool | ops cap_ops*2 | pair records cap_ops*4 | op links cap_ops/2 | entries cap_ent/2 | spectrum ch*half ]
// FAST: the stream shape guarantees the pair path for every residue and the fused tail (host: nvh_api.hip decides
// per stream); the general paths are then not even compiled in, which is worth registers and instruction cache.
// IMDCT (FAST only, block1 <= 2048): the inverse MDCT runs in the same workgroup, one wavefront per channel, straight
// from the LDS spectrum, and the work planes receive the compact IMDCT output k_ola_compact expects -- the
// spectrum never travels through HBM.
template <bool FLOOR0, bool FAST, bool IMDCT = false, int NT = SP_THREADS>
__device__ __forceinline__ void spectrum_body(const NvhDevSetup& S, const NvhDevBatch& Bt, float* __restrict__ work,
                                              int* __restrict__ err, int cap_pass, int cap_ops, int cap_ent, float* smem,
                                              long long* dbg = nullptr, int phase_mask = 15) {
  const int nch = S.channels;
  // channels whose floors are prepared concurrently (one wavefront each, one scratch block each) on the general path
  constexpr int grp = NT / 64;
  const int ngrp_lds = nch < grp ? nch : grp;
  float* s_db = smem;
  float* s_coeff = smem + 256;  // FLOOR0 only
  FloorScratch* fs = reinterpret_cast<FloorScratch*>(smem + (FLOOR0 ? 512 : 256));
  // 8-wavefront variant (never the fused tail): the floors are prepared after the residue and coupling phases, when the
  // staged side information is dead, so the floor scratch shares its storage
  constexpr bool kAliasScratch = NT > SP_THREADS;
  uint32_t* s_pass = reinterpret_cast<uint32_t*>(kAliasScratch ? fs : fs + ngrp_lds);  // per pass, 16 words: residue, op_begin[0..8] (frame relative), residue geometry (below)
  NvhDevBook* s_books = reinterpret_cast<NvhDevBook*>(s_pass + cap_pass * 16);
  uint32_t* s_lat = reinterpret_cast<uint32_t*>(reinterpret_cast<float*>(s_books) + S.nbooks * 8);
  NvhResOp* s_ops = reinterpret_cast<NvhResOp*>(s_lat + ((S.lattice_words + 3) & ~3));
  uint4* s_oprec = reinterpret_cast<uint4*>(reinterpret_cast<float*>(s_ops) + cap_ops * 2);  // pair-path op records
  uint16_t* s_link = reinterpret_cast<uint16_t*>(reinterpret_cast<float*>(s_oprec) + cap_ops * 4);  // cap_ops % 8 == 0
  uint16_t* s_ent = s_link + cap_ops;
  float* spec = reinterpret_cast<float*>(s_ent) + (cap_ent >> 1);  // [ch][half], 16-byte aligned (cap_ent % 8 == 0)
  if (kAliasScratch && spec < reinterpret_cast<float*>(fs + ngrp_lds)) spec = reinterpret_cast<float*>(fs + ngrp_lds);

  const int f = blockIdx.x;
#define DBG_T(k) do { if (dbg && threadIdx.x == 0) dbg[(long long)blockIdx.x * 24 + (k)] = clock64(); } while (0)
  DBG_T(0);
  if (phase_mask & 64) return;   // profiling builds: the launch itself (NVH_DEBUG_SPECTRUM_MASK)
  if (dbg && threadIdx.x == 0) {
    dbg[(long long)blockIdx.x * 24 + 22] = wall_clock64();
    dbg[(long long)blockIdx.x * 24 + 19] = ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);  // XCC_ID, HW_ID
  }
  // Every frame (drain pseudo-frames included) owns exactly `channels` channel records (host_parse.cpp), so the
  // channel records and the floor data behind them do not have to wait for the frame record.
  const int tid = threadIdx.x;
  const int wv = tid >> 6, lane = tid & 63;
  const NvhChan* chans = Bt.chans + (long long)f * nch;  // == fr.chan_off
  const NvhFrame fr = Bt.frames[f];
  // floor lane data of the first channel group: independent of the residue, so fetch it first
  const FloorLane first_lane = load_floor_lane(S, Bt, chans, wv, nch, lane);
  if (fr.n == 0) return;
  const int half = fr.n >> 1;
  const NvhDevMapping mp = S.mappings[fr.mapping];
  DBG_T(1);
  if (phase_mask & 128) return;  // profiling builds: launch + frame record + mapping record

  // ---- stage the frame's side information (16-byte copies), clear the spectrum, prepare the floors ----
  // Mono / stereo Floor1 streams take the fused tail: their posts are unwrapped here, one wavefront per channel,
  // while the remaining wavefronts do the staging -- two dependent-load chains side by side instead of in series.
  const bool fused_tail = FAST || (!FLOOR0 && S.fused_tail_ok);
  const int nprep = fused_tail ? nch : 0;  // wavefronts [0, nprep) prepare floors, the others stage
  const int npass = (int)(fr.pass_end - fr.pass_begin);
  // the host sizes the three capacities from the batch's largest frame (nvh_api.hip) and launches the unfused
  // kernels instead when that does not fit; a frame beyond them would be a host bug
  if ((int)fr.op_count > cap_ops || (int)fr.ent_count + 7 > cap_ent || npass > cap_pass) __builtin_trap();
  // the entry slice starts at any 2-byte offset: copy from the enclosing 16-byte boundary
  const unsigned ent_shift = fr.ent_begin & 7u;
  DBG_T(20);
  // both sources are indexed relative to the frame's slice (op.ent_off and pass->op_begin[] are batch offsets)
  const NvhResOp* ops = s_ops;
  const uint16_t* ent = s_ent + ent_shift;
  // pair path: everything a lane needs about an op and its codebook in one 16-byte record, resolved once per op
  // (from the staged copies) instead of once per element in the stage loops
  //   x: entry slice offset | first bin << 16      y: lattice pool offset | lat_values << 16
  //   z: ceil(2^32 / lat_values)                   w: dim | channel << 8 | ceil(2^16 / dim) << 16
  auto build_pair_records = [&](int first, int stride) {
    for (int ps = 0; ps < npass; ++ps) {
      const uint32_t* P = s_pass + ps * 16;
      const unsigned rflags = __builtin_amdgcn_readfirstlane(P[10]);
      if (!(rflags & 0x100u)) continue;
      const int o_end = __builtin_amdgcn_readfirstlane((int)P[1 + NVH_MAX_STAGES]);
      const unsigned rch = __builtin_amdgcn_readfirstlane(P[11]), psz = __builtin_amdgcn_readfirstlane(P[12]);
      const unsigned rch_magic = __builtin_amdgcn_readfirstlane(P[14]), rbegin = __builtin_amdgcn_readfirstlane(P[15]);
      for (int o = __builtin_amdgcn_readfirstlane((int)P[1]) + first; o < o_end; o += stride) {
        const NvhResOp op = ops[o];
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
  };
  // staging pieces; (first, stride) say which lanes of the staging wavefronts share a piece
  auto stage_ops = [&](int st, int sn) {
    const uint4* gb = reinterpret_cast<const uint4*>(S.books);
    for (int i = st; i < S.nbooks * 2; i += sn) reinterpret_cast<uint4*>(s_books)[i] = gb[i];
    const uint2* go = reinterpret_cast<const uint2*>(Bt.ops + fr.op_begin);
    for (int i = st; i < (int)fr.op_count; i += sn) reinterpret_cast<uint2*>(s_ops)[i] = go[i];
    if (FAST) {
      const uint16_t* gl = Bt.op_link + fr.op_begin;
      for (int i = st; i < (int)fr.op_count; i += sn) s_link[i] = gl[i];
    }
    // pass records: op ranges per stage plus the residue geometry the stage loops need, so that those do not
    // start with another global round trip (frame -> pass -> residue)
    for (int p = st; p < npass; p += sn) {
      const NvhResPass* gp = Bt.passes + fr.pass_begin + p;
      const NvhDevResidue* Rp = &S.residues[gp->residue];
      uint32_t* P = s_pass + p * 16;
      P[0] = (uint32_t)gp->residue;
      for (int k = 0; k <= NVH_MAX_STAGES; ++k) P[1 + k] = gp->op_begin[k] - fr.op_begin;
      P[10] = (uint32_t)Rp->type | (Rp->pair_path ? 0x100u : 0u) | (Rp->sequential ? 0x200u : 0u) | (Rp->fast ? 0x400u : 0u);
      P[11] = (uint32_t)Rp->real_channels;
      P[12] = (uint32_t)Rp->partition_size;
      P[13] = Rp->hp_magic;
      P[14] = Rp->rch_magic;
      P[15] = (uint32_t)Rp->begin;
    }
  };
  auto stage_rest = [&](int st, int sn) {
    for (int i = st; i < 256; i += sn) s_db[i] = k_inverse_db[i];
    for (int i = st; i < S.lattice_words; i += sn) s_lat[i] = S.lattice[i];
    const uint4* ge = reinterpret_cast<const uint4*>(Bt.entries + (fr.ent_begin - ent_shift));
    const int nvec = (int)((ent_shift + fr.ent_count + 7u) >> 3);
    for (int i = st; i < nvec; i += sn) reinterpret_cast<uint4*>(s_ent)[i] = ge[i];
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int i = st; i < (nch * half) >> 2; i += sn) reinterpret_cast<float4*>(spec)[i] = z;  // Mapping.cs:108
  };
  if (wv < nprep) {
    if (phase_mask & 4) floor_prepare(&fs[wv], first_lane, lane, half, err, S.recip);
  } else if (FAST) {
    // The floor unwrap is the longer chain, so the staging wavefronts have time left: the last of them stages the op
    // side and turns it into pair records on its own (wavefront-local ordering only), the other(s) take the rest;
    // when the floors are ready the residue sweep can start at once.
    const int nstage = NT / 64 - nprep, sw = wv - nprep;  // nch <= 2: two or three staging wavefronts
    if (sw == nstage - 1) {
      stage_ops(lane, 64);
      sp_wave_sync();
      if (!(phase_mask & 256)) build_pair_records(lane, 64);
      // Chain heads of every pass, compacted (op indices, 16 bits each) over the staged ops, which nothing reads any more:
      // the sweep below walks one chain per lane and pair of bins, and two thirds of a three-stage pass's ops are not
      // heads -- whole wavefronts of the sweep used to find nothing but "not a head".  P[0] becomes the pass's head range.
      sp_wave_sync();
      uint16_t* heads = reinterpret_cast<uint16_t*>(s_ops);
      int hcount = 0;
      for (int ps = 0; ps < npass; ++ps) {
        uint32_t* P = s_pass + ps * 16;
        const int o_lo = __builtin_amdgcn_readfirstlane((int)P[1]), o_hi = __builtin_amdgcn_readfirstlane((int)P[1 + NVH_MAX_STAGES]);
        const int hb = hcount;
        for (int base = o_lo; base < o_hi; base += 64) {
          const int o = base + lane;
          const bool head = o < o_hi && !(s_link[o] & 0x8000u);
          const unsigned long long m = __ballot(head);
          if (head) heads[hcount + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)o;
          hcount += __popcll(m);
        }
        if (lane == 0) P[0] = (uint32_t)hb | ((uint32_t)hcount << 16);
      }
    } else {
      stage_rest(sw * 64 + lane, (nstage - 1) * 64);
    }
  } else {
    const int st = tid - nprep * 64, sn = NT - nprep * 64;
    stage_ops(st, sn);
    stage_rest(st, sn);
  }
  DBG_T(21);
  __syncthreads();
  if (!FAST) {
    build_pair_records(tid, NT);
    __syncthreads();
  }
  DBG_T(2);

  // Inverse coupling of a stereo frame (Mapping.cs:137-182) needs both channels of a bin after ALL residue adds.  When the
  // frame's only residue pass is a two-channel Residue2, the pair lane that finishes a partition's chain holds exactly
  // that -- bin x of channel 0 and of channel 1 -- in registers: it couples before its one write (bins no chain covers
  // are still (0, 0), which the coupling maps to itself).  The tail is then independent per channel.
  int cpl_mg = 0;
  bool couple = false;
  if (fused_tail && nch == 2 && mp.coupling_steps == 1) {
    cpl_mg = S.coupling[mp.coupling_off];  // the angle channel is the other one
    couple = (fr.exec_mask & 3u) != 0;
  }
  const bool sweep_couples = FAST && couple && npass == 1 && (__builtin_amdgcn_readfirstlane(s_pass[10]) & 0xFFu) == 2 &&
                             __builtin_amdgcn_readfirstlane(s_pass[11]) == 2;

  // ---- residue ----
  for (int ps = 0; (phase_mask & 1) && ps < npass; ++ps) {
    // values read back from LDS are wave-uniform: say so, or every use downstream turns into vector code
    const uint32_t* P = s_pass + ps * 16;
    const unsigned rflags = __builtin_amdgcn_readfirstlane(P[10]);
    const int rtype = (int)(rflags & 0xFFu);
    const int psize = __builtin_amdgcn_readfirstlane((int)P[12]);
    const unsigned rch = __builtin_amdgcn_readfirstlane(P[11]);
    const unsigned hp_magic = __builtin_amdgcn_readfirstlane(P[13]), rch_magic = __builtin_amdgcn_readfirstlane(P[14]);
    const NvhDevResidue* Rg = FAST ? nullptr : &S.residues[__builtin_amdgcn_readfirstlane((int)P[0])];  // general paths only
    long long t_prev = dbg ? clock64() : 0;
    if (dbg && threadIdx.x == 0) dbg[(long long)blockIdx.x * 24 + 7] = t_prev;
    if (FAST) {
      // All stages of the pass in one sweep.  The host links the ops that add to the same partition/channel across
      // stages (op_link); a lane takes one pair of bins of one chain *head*, walks the chain in stage order with
      // the running sums in registers, and writes once.  Same additions in the same order as the reference's
      // stage loop, without a barrier and an LDS read-modify-write per stage.
      const unsigned hp = (unsigned)psize >> 1;
      const unsigned hrange = __builtin_amdgcn_readfirstlane(P[0]);  // chain heads of this pass (phase A)
      const uint16_t* heads = reinterpret_cast<const uint16_t*>(s_ops) + (hrange & 0xFFFFu);
      const unsigned total = ((hrange >> 16) - (hrange & 0xFFFFu)) * hp;
      for (unsigned idx = tid; idx < total; idx += NT) {
        const unsigned oq = hp > 1 ? __umulhi(idx, hp_magic) : idx;
        const unsigned i2 = idx - oq * hp, i = i2 << 1;  // pair / first component index inside the partition
        unsigned o = heads[oq];
        unsigned link = s_link[o];
        uint4 rec = s_oprec[o];
        const unsigned xbase = rec.x >> 16;
        unsigned c0, x0, c1, x1;
        if (rtype == 1 || rch == 1) {
          c0 = c1 = (rec.w >> 8) & 0xFFu;
          x0 = xbase + i;
          x1 = x0 + 1;
        } else if (rch == 2) {
          c0 = 0; c1 = 1;
          x0 = x1 = xbase + i2;
        } else {
          const unsigned qi = __umulhi(i, rch_magic);
          c0 = i - qi * rch;
          x0 = xbase + qi;
          c1 = c0 + 1; x1 = x0;
          if (c1 == rch) { c1 = 0; ++x1; }
        }
        const bool in0 = x0 < (unsigned)half, in1 = x1 < (unsigned)half;
        float* p0 = spec + c0 * (unsigned)half + x0;
        float* p1 = spec + c1 * (unsigned)half + x1;
        float a0 = in0 ? *p0 : 0.0f, a1 = in1 ? *p1 : 0.0f;
        for (;;) {
          const unsigned dims = rec.w & 0xFFu, lv = rec.y >> 16;
          const unsigned j = (i * (rec.w >> 16)) >> 16;  // i / dims (i < 4096, dims <= 16: exact)
          const unsigned comp = i - j * dims;
          unsigned q = ent[(rec.x & 0xFFFFu) + j];
          if (q != NVH_ENTRY_SKIP) {
            const uint32_t* lat = s_lat + (rec.y & 0xFFFFu);
            if (comp) q = __umulhi(q, lat[lv + comp]);  // e / lv^comp
            // two base-lv digits (lv == 1: the magic is 0 and so are q and both digits)
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
        if (sweep_couples) {  // rtype == 2, rch == 2: a0 / a1 are bin x0 of channel 0 / 1
          if (cpl_mg == 0) couple1(a0, a1); else couple1(a1, a0);
        }
        if (in0) *p0 = a0;
        if (in1) *p1 = a1;
      }
      __syncthreads();
      continue;
    }
#pragma unroll 1
    for (int s = 0; s < NVH_MAX_STAGES; ++s) {
      const unsigned ob = __builtin_amdgcn_readfirstlane(P[1 + s]);
      const unsigned oe = __builtin_amdgcn_readfirstlane(P[2 + s]);
      if (ob == oe) continue;
      if (rflags & 0x100u) {
        // every book of this residue is a lattice of even dimension: one lane adds two consecutive components of
        // one codebook entry (for stereo type 2 that is one bin of both channels).  The VQ lookup is two base-
        // lat_values digits of the entry number, peeled with exact reciprocal multiplies; nothing leaves LDS.
        const unsigned hp = (unsigned)psize >> 1;
        const unsigned total = (oe - ob) * hp;
        for (unsigned idx = tid; idx < total; idx += NT) {
          const unsigned o = hp > 1 ? __umulhi(idx, hp_magic) : idx;
          const unsigned i2 = idx - o * hp, i = i2 << 1;  // pair / first component index inside the partition
          const uint4 rec = s_oprec[ob + o];
          const unsigned dims = rec.w & 0xFFu, lv = rec.y >> 16;
          const unsigned j = (i * (rec.w >> 16)) >> 16;  // i / dims (i < 4096, dims <= 16: exact)
          const unsigned comp = i - j * dims;
          unsigned q = ent[(rec.x & 0xFFFFu) + j];
          if (q == NVH_ENTRY_SKIP) continue;
          const uint32_t* lat = s_lat + (rec.y & 0xFFFFu);
          if (comp) q = __umulhi(q, lat[lv + comp]);  // e / lv^comp
          // two base-lv digits (lv == 1: the magic is 0 and so are q and both digits)
          const unsigned q1 = __umulhi(q, rec.z);
          const unsigned d0 = q - q1 * lv;
          const unsigned d1 = q1 - __umulhi(q1, rec.z) * lv;
          const float v0 = __uint_as_float(lat[d0]), v1 = __uint_as_float(lat[d1]);
          const unsigned xbase = rec.x >> 16;
          unsigned c0, x0, c1, x1;
          if (rtype == 1 || rch == 1) {
            c0 = c1 = (rec.w >> 8) & 0xFFu;
            x0 = xbase + i;
            x1 = x0 + 1;
          } else if (rch == 2) {
            c0 = 0; c1 = 1;
            x0 = x1 = xbase + i2;
          } else {
            const unsigned qi = __umulhi(i, rch_magic);
            c0 = i - qi * rch;
            x0 = xbase + qi;
            c1 = c0 + 1; x1 = x0;
            if (c1 == rch) { c1 = 0; ++x1; }
          }
          if (x0 < (unsigned)half) {
            float* p = spec + c0 * (unsigned)half + x0;
            *p = *p + v0;
          }
          if (x1 < (unsigned)half) {
            float* p = spec + c1 * (unsigned)half + x1;
            *p = *p + v1;
          }
        }
        __syncthreads();
      } else if ((rflags & 0x600u) == 0x400u) {  // fast, not sequential
        const NvhDevResidue R = *Rg;
        // elements of one stage never alias (that is what !sequential means), so four of them are fetched as
        // independent dependency chains before their adds are committed
        const int total = (int)(oe - ob) * psize;
        for (int base = tid; base < total; base += 4 * NT) {
          float* tp[4];
          float tv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int idx = base + u * NT;
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
      } else if (!(rflags & 0x200u)) {
        const NvhDevResidue R = *Rg;
        const int total = (int)(oe - ob) * psize;
        for (int idx = tid; idx < total; idx += NT) {
          unsigned o = (unsigned)idx / (unsigned)psize;
          int i = idx - (int)o * psize;
          const NvhResOp op = ops[ob + o];
          residue_apply_lds(s_books[op.book], S.vq, R, op, ent, fr.ent_begin, i, spec, half);
        }
        __syncthreads();
      } else {
        // partitions may alias (quirk B-1 / vector overrun): keep the reference's partition order
        const NvhDevResidue R = *Rg;
        for (unsigned o = ob; o < oe; ++o) {
          const NvhResOp op = ops[o];
          const NvhDevBook bk = s_books[op.book];
          const int dims = (int)bk.dim;
          const int cnt = ((psize + dims - 1) / dims) * dims;
          for (int i = tid; i < cnt; i += NT) residue_apply_lds(bk, S.vq, R, op, ent, fr.ent_begin, i, spec, half);
          __syncthreads();
        }
      }
      if (dbg) {
        long long t_now = clock64();
        if (threadIdx.x == 0) {
          dbg[(long long)blockIdx.x * 24 + 8 + s] = t_now - t_prev;
          dbg[(long long)blockIdx.x * 24 + 16 + s] = (long long)(oe - ob);
        }
        t_prev = t_now;
      }
    }
  }

  DBG_T(3);
  float* planes = work + (long long)f * nch * S.block1;

  if (fused_tail && !(phase_mask & 2)) return;  // profiling aid (NVH_DEBUG_SPECTRUM_MASK)
  if (fused_tail) {
    // ---- fused tail: inverse coupling (Mapping.cs:137-182), floor apply and the store, 4 bins per lane ----
    const int mg = cpl_mg;
    const bool tail_couples = couple && !sweep_couples;
    const int md0 = __builtin_amdgcn_readfirstlane(fs[0].mode), md1 = nch == 2 ? __builtin_amdgcn_readfirstlane(fs[1].mode) : 0;
    // TB bins per lane: the segment search and the recurrence restart are paid once per TB bins and channel
    constexpr int TB = SP_TAIL_BINS;
    if (nch == 2 && !tail_couples && NT == SP_THREADS) {
      // Channels are independent here: half of the workgroup per channel, 8 consecutive bins per lane -- the segment
      // search and the restart of the error recurrence are paid once per 8 bins instead of once per 4.
      constexpr int TS = 8;
      const int c = tid >> 7;  // wave-uniform
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
          for (int q = 0; q < TS; q += 4) {
            if (IMDCT) *reinterpret_cast<float4*>(sp + x0 + q) = *reinterpret_cast<float4*>(r + q);
            else *reinterpret_cast<float4*>(planes + (long long)c * S.block1 + x0 + q) = *reinterpret_cast<float4*>(r + q);
          }
        }
      } else if (!IMDCT) {  // the channel does not execute: its residue goes out as it is (quirk B-4)
        for (int x0 = (tid & 127) * 4; x0 < half; x0 += 128 * 4)
          *reinterpret_cast<float4*>(planes + (long long)c * S.block1 + x0) = *reinterpret_cast<const float4*>(sp + x0);
      }
    } else
    for (int x0 = tid * TB; x0 < half; x0 += NT * TB) {
      float r0[TB], r1[TB];
#pragma unroll
      for (int q = 0; q < TB; q += 4) {
        *reinterpret_cast<float4*>(r0 + q) = *reinterpret_cast<const float4*>(spec + x0 + q);
        if (nch == 2) *reinterpret_cast<float4*>(r1 + q) = *reinterpret_cast<const float4*>(spec + half + x0 + q);
      }
      if (tail_couples) {
#pragma unroll
        for (int q = 0; q < TB; ++q) {
          if (mg == 0) couple1(r0[q], r1[q]); else couple1(r1[q], r0[q]);
        }
      }
      float m[TB];
      if (md0 == 1) {
        floor_walk<TB>(&fs[0], s_db, x0, m);
#pragma unroll
        for (int q = 0; q < TB; ++q) r0[q] = r0[q] * m[q];
      } else if (md0 == 2) {
#pragma unroll
        for (int q = 0; q < TB; ++q) r0[q] = 0.0f;  // Floor1.cs:218-221
      }
#pragma unroll
      for (int q = 0; q < TB; q += 4) {
        if (IMDCT) *reinterpret_cast<float4*>(spec + x0 + q) = *reinterpret_cast<float4*>(r0 + q);
        else *reinterpret_cast<float4*>(planes + x0 + q) = *reinterpret_cast<float4*>(r0 + q);
      }
      if (nch == 2) {
        if (md1 == 1) {
          floor_walk<TB>(&fs[1], s_db, x0, m);
#pragma unroll
          for (int q = 0; q < TB; ++q) r1[q] = r1[q] * m[q];
        } else if (md1 == 2) {
#pragma unroll
          for (int q = 0; q < TB; ++q) r1[q] = 0.0f;
        }
#pragma unroll
        for (int q = 0; q < TB; q += 4) {
          if (IMDCT) *reinterpret_cast<float4*>(spec + half + x0 + q) = *reinterpret_cast<float4*>(r1 + q);
          else *reinterpret_cast<float4*>(planes + S.block1 + x0 + q) = *reinterpret_cast<float4*>(r1 + q);
        }
      }
    }
    DBG_T(4);
    if (IMDCT && (phase_mask & 8)) {
      // ---- inverse MDCT (Mdct.cs:65-313), one wavefront per channel ----
      // The transform's LDS slice (n/2 floats + n/16 of padding) overlays the channel's own spectrum, which the
      // wavefront has fully in registers before its first store (imdct_wave_fast loads everything up front):
      // channel nch-1 spills its padding past the end of the spectrum area, the one before it into the (dead)
      // staging area in front of it.
      // The workgroup barrier between the floor multiply and the transform sits INSIDE imdct_wave<.., PRESYNC> for the
      // wavefronts that transform (behind their first table loads), and here for the others: one barrier each.
      if (wv < nch && ((fr.exec_mask >> wv) & 1u)) {
        const float* X = spec + wv * half;
        float* out = planes + (long long)wv * S.block1;
        float* scratch = spec + wv * half - (nch - 1 - wv) * (fr.n >> 4);
        const int sl = fr.mdct_slot;
        const float* A = S.mdct_a[sl];
        const float* B = S.mdct_b[sl];
        const float* C = S.mdct_c[sl];
        const float* TW = S.mdct_tw[sl];
        long long* stamp = (dbg && wv == 0) ? dbg + (long long)blockIdx.x * 24 + 8 : nullptr;  // profiling builds
        const int skip = (phase_mask >> 4) & 3;
        switch (fr.n) {
          case 256: imdct_wave<8, false, true, true, false, true>(X, out, nullptr, scratch, A, B, C, TW, lane, stamp, skip); break;
          case 512: imdct_wave<9, false, true, true, false, true>(X, out, nullptr, scratch, A, B, C, TW, lane, stamp, skip); break;
          case 1024: imdct_wave<10, false, true, true, false, true>(X, out, nullptr, scratch, A, B, C, TW, lane, stamp, skip); break;
          case 2048: imdct_wave<11, false, true, true, false, true>(X, out, nullptr, scratch, A, B, C, TW, lane, stamp, skip); break;
          default: __builtin_trap();  // host launches this kernel for 256 <= block0, block1 <= 2048 only
        }
      } else {
        __syncthreads();
        if (wv < nch) {
          // Mapping.cs:192-196: the residue stays in [0, n/2) (k_ola_compact windows it); its tail quarter is zero
          const float* X = spec + wv * half;
          float* out = planes + (long long)wv * S.block1;
          for (int i = lane * 4; i < half; i += 256) *reinterpret_cast<float4*>(out + i) = *reinterpret_cast<const float4*>(X + i);
          for (int i = lane * 4; i < (half >> 1); i += 256) *reinterpret_cast<float4*>(out + half + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
      }
    }
    DBG_T(5);
    DBG_T(6);
    if (dbg && threadIdx.x == 0) dbg[(long long)blockIdx.x * 24 + 23] = wall_clock64();
    return;
  }
  if (FAST) return;

  // ---- general tail: any channel count, any number of coupling steps, Floor0 ----
  // inverse coupling, last step first (Mapping.cs:137-182)
  for (int st = mp.coupling_steps - 1; st >= 0; --st) {
    const int mg = S.coupling[mp.coupling_off + 2 * st], an = S.coupling[mp.coupling_off + 2 * st + 1];
    if (chans[an].exec || chans[mg].exec) {
      float* M = spec + mg * half;
      float* Aa = spec + an * half;
      for (int j = tid; j < half; j += NT) {
        float vm = M[j], va = Aa[j];
        couple1(vm, va);
        M[j] = vm;
        Aa[j] = va;
      }
    }
    __syncthreads();
  }

  DBG_T(4);
  // floors, `grp` channels at a time: wavefront w < grp prepares channel c0 + w
  for (int c0 = 0; c0 < nch; c0 += grp) {
    const FloorLane fl_lane = (c0 == 0) ? first_lane : load_floor_lane(S, Bt, chans, c0 + wv, nch, lane);
    if (c0 + wv < nch) floor_prepare(&fs[wv], fl_lane, lane, half, err, S.recip);
    __syncthreads();

    // render / apply: all threads over the group's channels
    const int ngrp = (nch - c0) < grp ? (nch - c0) : grp;
    for (int k = 0; k < ngrp; ++k) {
      const int cc = c0 + k;
      const int md = __builtin_amdgcn_readfirstlane(fs[k].mode);
      float* res = spec + cc * half;
      if (md == 0) continue;
      if (md == 2) {
        for (int i = tid; i < half; i += NT) res[i] = 0.0f;  // Floor1.cs:218-221 / Floor0.cs:208-211
        continue;
      }
      if (md == 1) {
        for (int x0 = tid * 4; x0 < half; x0 += NT * 4) {
          float m[4];
          floor_walk<4>(&fs[k], s_db, x0, m);
          float4 v = *reinterpret_cast<float4*>(res + x0);
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
        floor0_curve<NT>(S, &S.floors[ck.floor].f0, Bt.coeffs + ck.data_off, ck.amp, fr.mdct_slot, res, half, s_coeff, tid, err);
      }
    }
    __syncthreads();
  }

  DBG_T(5);
  if (IMDCT) {
    // ---- inverse MDCT behind the general path (k_spectrum_gen8_imdct): one wavefront per channel (NT / 64 >= channels) ----
    // Every wavefront first takes its channel's whole spectrum into registers; behind the workgroup barrier inside
    // imdct_wave<.., WGSYNC> the floor scratch, the staged side information and all spectra are dead, and the transforms'
    // slices (n/2 + n/16 floats each) are laid out back to back from the start of that area.
    float* slices = smem + 256;
    const bool mine = wv < nch;
    if (mine && chans[wv].exec) {
      const float* X = spec + wv * half;
      float* out = planes + (long long)wv * S.block1;
      float* scratch = slices + wv * (half + (fr.n >> 4));
      const int sl = fr.mdct_slot;
      const float* A = S.mdct_a[sl];
      const float* B = S.mdct_b[sl];
      const float* C = S.mdct_c[sl];
      const float* TW = S.mdct_tw[sl];
      switch (fr.n) {
        case 256: imdct_wave<8, false, true, true, true>(X, out, nullptr, scratch, A, B, C, TW, lane); break;
        case 512: imdct_wave<9, false, true, true, true>(X, out, nullptr, scratch, A, B, C, TW, lane); break;
        case 1024: imdct_wave<10, false, true, true, true>(X, out, nullptr, scratch, A, B, C, TW, lane); break;
        case 2048: imdct_wave<11, false, true, true, true>(X, out, nullptr, scratch, A, B, C, TW, lane); break;
        case 4096: imdct_wave<12, false, true, true, true>(X, out, nullptr, scratch, A, B, C, TW, lane); break;
        default: __builtin_trap();  // the host launches this kernel for block sizes up to 4096 only
      }
    } else {
      if (mine) {
        // Mapping.cs:192-196: the residue stays in [0, n/2) (k_ola_compact windows it); its tail quarter is zero
        const float* X = spec + wv * half;
        float* out = planes + (long long)wv * S.block1;
        for (int i = lane * 4; i < half; i += 256) *reinterpret_cast<float4*>(out + i) = *reinterpret_cast<const float4*>(X + i);
        for (int i = lane * 4; i < (half >> 1); i += 256) *reinterpret_cast<float4*>(out + half + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
      __syncthreads();  // the one barrier every transforming wavefront passes
    }
    DBG_T(6);
    return;
  }
  // spectrum -> work planes
  const int q4 = half >> 2;
  for (int i = tid; i < nch * q4; i += NT) {
    int c = i / q4, k = i - c * q4;
    reinterpret_cast<float4*>(planes + (long long)c * S.block1)[k] = reinterpret_cast<const float4*>(spec + c * half)[k];
  }
  DBG_T(6);
}

// 8 waves per SIMD = 8 resident workgroups per CU: the register budget (64 VGPRs, 96 SGPRs) is part of the design
extern "C" __global__ void __launch_bounds__(SP_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_spectrum(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err, int cap_pass, int cap_ops,
           int cap_ent NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  spectrum_body<false, true>(S, Bt, work, err, cap_pass, cap_ops, cap_ent, smem NVH_DBG_ARGS);
}

// k_spectrum with the inverse MDCT behind it (block sizes 256..2048): writes the compact IMDCT output.
extern "C" __global__ void __launch_bounds__(SP_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_spectrum_imdct(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err, int cap_pass, int cap_ops,
                 int cap_ent NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  spectrum_body<false, true, true>(S, Bt, work, err, cap_pass, cap_ops, cap_ent, smem NVH_DBG_ARGS);
}

// Any other Floor1 stream shape (more than two channels, several coupling steps, non-lattice books, aliasing
// partitions, Residue0).
extern "C" __global__ void __launch_bounds__(SP_THREADS)
k_spectrum_gen(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err, int cap_pass, int cap_ops,
               int cap_ent NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  spectrum_body<false, false>(S, Bt, work, err, cap_pass, cap_ops, cap_ent, smem NVH_DBG_ARGS);
}

// k_spectrum_gen with 8 wavefronts per workgroup, for more than four channels: 48 KB of spectrum (six channels, n = 4096)
// leave 3 workgroups per CU, so the wavefronts have to come from inside the workgroup; all floors are unwrapped at once.
extern "C" __global__ void __launch_bounds__(512)
k_spectrum_gen8(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err, int cap_pass, int cap_ops,
                int cap_ent NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  spectrum_body<false, false, false, 512>(S, Bt, work, err, cap_pass, cap_ops, cap_ent, smem NVH_DBG_ARGS);
}

// k_spectrum_gen with the inverse MDCT behind it: the fused-tail form for mono / stereo (block sizes 256..2048, as in
// k_spectrum_imdct), the general form for three and four channels (256..4096).
extern "C" __global__ void __launch_bounds__(SP_THREADS)
k_spectrum_gen_imdct(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err, int cap_pass, int cap_ops,
                     int cap_ent NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  spectrum_body<false, false, true>(S, Bt, work, err, cap_pass, cap_ops, cap_ent, smem NVH_DBG_ARGS);
}

// k_spectrum_gen8 with the inverse MDCT behind it (block sizes 256..4096, at most 8 channels): writes the compact IMDCT output.
extern "C" __global__ void __launch_bounds__(512)
k_spectrum_gen8_imdct(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err, int cap_pass, int cap_ops,
                      int cap_ent NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  spectrum_body<false, false, true, 512>(S, Bt, work, err, cap_pass, cap_ops, cap_ent, smem NVH_DBG_ARGS);
}

// Variant for setups that contain a Floor0 (double-precision cos / sqrt / exp: costs registers, kept apart).
extern "C" __global__ void __launch_bounds__(SP_THREADS)
k_spectrum_f0(NvhDevSetup S, NvhDevBatch Bt, float* __restrict__ work, int* __restrict__ err, int cap_pass, int cap_ops,
              int cap_ent) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  spectrum_body<true, false>(S, Bt, work, err, cap_pass, cap_ops, cap_ent, smem);
}

// IFloor.Apply for Floor1 (Floor1.cs:186-222) as an operator on its own: one wavefront per item unwraps the raw
// posts an Unpack call produced and multiplies the item's n/2 residue values by the curve (or clears them when
// the floor carried no energy, :218-221).  Same device functions as the fused tail of the spectrum kernels.
// status[item] receives the NVH_DEVERR_* bits of that item alone.
extern "C" __global__ void __launch_bounds__(64)
k_floor1_apply(NvhDevSetup S, int floor_idx, const uint16_t* __restrict__ posts, const int32_t* __restrict__ counts, int n,
               float* __restrict__ data, long long stride, int* __restrict__ status) {
  __shared__ FloorScratch Q;
  __shared__ float s_db[256];
  const int item = (int)blockIdx.x, lane = (int)threadIdx.x, half = n >> 1;
  for (int i = lane; i < 256; i += 64) s_db[i] = k_inverse_db[i];
  const NvhDevFloor1* F = &S.floors[floor_idx].f1;
  FloorLane L;
  L.pc = counts[item];
  L.mode = L.pc > 0 ? 1 : 2;
  L.levels = F->levels;
  L.range = F->range;
  L.mult = F->multiplier;
  L.level = 0; L.lo = 0; L.hi = 1; L.x = 0; L.x_lo = 0; L.x_hi = 1; L.val = 0; L.sorted = 0; L.x_sorted = 0; L.adx_magic = 0;
  if (lane < L.pc) {
    L.lo = F->l_neigh[lane];
    L.hi = F->h_neigh[lane];
    L.level = F->level[lane];
    L.x = F->x_list[lane];
    L.val = posts[(long long)item * NVH_MAX_POSTS + lane];
    L.sorted = F->sort_idx[lane];
    L.x_lo = F->x_lo[lane];
    L.x_hi = F->x_hi[lane];
    L.x_sorted = F->x_sorted[lane];
    L.adx_magic = F->adx_magic[lane];
  }
  floor_prepare(&Q, L, lane, half, status + item, S.recip);
  __syncthreads();
  float* res = data + (long long)item * stride;
  if (L.mode == 2) {
    for (int i = lane; i < half; i += 64) res[i] = 0.0f;
    return;
  }
  for (int x0 = lane * 4; x0 < half; x0 += 64 * 4) {
    float m[4];
    floor_walk<4>(&Q, s_db, x0, m);
    float4 v = *reinterpret_cast<float4*>(res + x0);
    v.x = v.x * m[0];
    v.y = v.y * m[1];
    v.z = v.z * m[2];
    v.w = v.w * m[3];
    *reinterpret_cast<float4*>(res + x0) = v;
  }
}

// IFloor.Apply for Floor0 (Floor0.cs:152-212) as an operator: item b scales data[b*stride .. +n/2) by the LSP curve of its
// coefficients, or clears it when its amplitude is not positive (:208-211).  The curve's value per Bark section comes from the
// host (host_slab.cpp: floor0_section_values -- the reference's expression shapes evaluated with the host's libm, the same the
// CPU oracle uses); here the gather by barkMap[i] and the multiply.  skip[b] != 0: the reference would throw, nothing is done.
extern "C" __global__ void __launch_bounds__(SP_THREADS)
k_floor0_apply(const int32_t* __restrict__ bark, const float* __restrict__ qk, int K, const float* __restrict__ amps,
               const int32_t* __restrict__ skip, int n, float* __restrict__ data, long long stride) {
  const int item = (int)blockIdx.x, tid = (int)threadIdx.x, half = n >> 1;
  if (skip[item]) return;
  float* res = data + (long long)item * stride;
  if (!(amps[item] > 0.0f)) {
    for (int i = tid; i < half; i += SP_THREADS) res[i] = 0.0f;
    return;
  }
  const float* q = qk + (long long)item * K;
  for (int i = tid; i < half; i += SP_THREADS) res[i] = res[i] * q[bark[i]];
}
