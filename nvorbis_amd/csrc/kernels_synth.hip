// kernels_synth.hip -- the slab synthesis kernels (round 3): residue adds + inverse coupling + Floor1 multiply + inverse MDCT
// of one frame per workgroup, fed by ONE LDS-DMA round trip.  k_synth: mono / stereo, blocks up to 2048; k_synth8: up to eight
// channels, blocks up to 8192.  They are the decode path of every batch inside their contract (nvh_launch.hip: slab_path); their
// input -- one slab per frame -- is written by the packet parsers themselves: host_slab.cpp on the host parser's thread,
// kernels_parse.hip (parse_body<.., SLAB>) in GPU-parse mode.
//
//   Array.Clear + IResidue.Decode adds   Mapping.cs:108,133; Residue1.cs:8-26, Residue2.cs:23-47
//   inverse square-polar coupling         Mapping.cs:137-182
//   IFloor.Apply (the multiply)           Floor1.cs:196-222, RenderLineMulti :316-341, inverse_dB_table :345-410
//   IMdct.Reverse                         Mdct.cs:65-313 (imdct_wave.h)
//   (integer side: the packet parsers)    Floor1.UnwrapPosts :224-297, the sorted / flagged post walk :196-216,
//                                         the (stage, partition, channel) geometry of Residue0.cs:157-170, Residue2.cs:23-47
//
// Why a second form of k_spectrum_imdct (kernels_spectrum.hip), same arithmetic (stream shapes: Floor1, lattice books of even
// dimension, no aliasing partitions, one residue pass per frame, <= 8 channels, blocks 256..4096): that kernel's workgroup spent 10.8 k of its
// 29.8 k cycles before its first useful instruction -- frame record, then the slices the record points to, then the setup
// records those point to (three dependent global round trips), then copies of all of it into LDS through registers, pair
// records, chain-head compaction, and the Floor1 unwrap (a chain of dependent LDS round trips on two otherwise idle
// wavefronts).  None of that depends on a float.  It is done by the packet parsers (host_slab.cpp; kernels_parse.hip in slab mode;
// integer work only), which leaves every frame's side information as one contiguous slab in its final LDS layout
// (nvh_format.h: NvhSlabHdr), at a fixed stride:
//   * the workgroup issues the slab's first 4 KB (global_load_lds_dwordx4, 1 KB per wavefront-instruction) and the
//     stream constants (inverse_dB_table + lattice pool) before it knows anything about the frame, clears the spectrum
//     while they fly, and passes one barrier: ONE memory round trip, no staging instructions, no VGPR round trip;
//   * the header arrives by a scalar load issued next to the DMA; a slab beyond 4 KB has its rest fetched in front of that
//     same barrier;
//   * the residue walk follows chain-major records (one ds_read_b128 per cascade stage, no link array); a lane owns eight
//     consecutive vector components of a chain through all cascade stages and, for mono / stereo, multiplies them by the floor
//     curve (segment list and per-four-bins segment table straight from the slab) before its one store.
// The transform is k_spectrum_imdct's (imdct_wave.h), the output the compact form k_ola_compact reads -- or, for mono / stereo
// batches in the steady state of a stream, no plane at all: PAIRED EMISSION (synth_emit below; nvh_format.h: NVH_EMIT_*).  The
// odd frames of a batch are synthesised first and leave their planes; the even frames follow in a second launch, keep their
// transform output in registers and do the window / overlap-add / clip / interleave (Mode.cs:160-166, StreamDecoder.cs:532-541,
// :391-415, Utils.cs:30-43) of both overlaps they take part in, with the neighbours' quarters fetched by LDS-DMA from the odd
// planes: half the planes are never written, none is read by a second kernel pass, and k_ola_compact runs only over the frames
// outside the steady state (block-size switches, the first frame of a batch, the block that becomes the carried tail).
// Bit-exactness: the additions of a partition happen in stage order inside the owning lane, the overlap-add is ola_sym's
// arithmetic (kernels.hip), every float expression is one rounded operation (-ffp-contract=off).
#include <hip/hip_runtime.h>

#include "imdct_wave.h"
#include "kernels_common.h"
#include "spectrum_dev.h"

namespace {

// Ablation builds (tools/build_variant.py NAME -DNVH_ABL_...): one phase of the slab kernels left out, to read its marginal cost
// in the regime the headline runs in (three streams, slots always full) -- results are wrong by construction, never shipped.
// k_synth's transforms with every pass's twiddles in registers a pass ahead (imdct_wave.h: PassesPF): the kernels have the registers
// for it since the output stage's addresses come from a table (round 5; k_synth_emit spills six registers and still comes out ahead).
// Same box, three streams: neither 193.8, the odd launch only 193.9, both 195.3 M frames/s.  NVH_NO_SYNTH_PF: build variants.
#ifdef NVH_NO_SYNTH_PF
constexpr bool kSynthPF = false, kSynthPFEmit = false;
#else
constexpr bool kSynthPF = true, kSynthPFEmit = true;
#endif
#ifdef NVH_ABL_NO_XFORM
constexpr int kAblSkip = 3;  // the transform without its radix passes and the D = 4, 2, 1 pass
#else
constexpr int kAblSkip = 0;
#endif

// 16 bytes per lane, 1 KB per wavefront-instruction, global -> LDS without a register round trip.  The LDS destination is
// wave-uniform (M0) + lane * 16; lanes that are switched off move nothing.
template <int AUX = 0>  // AUX 2: `nt`, for data this launch reads exactly once (the slabs)
__device__ __forceinline__ void dma16(const uint4* __restrict__ gsrc, float* lds_chunk_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_chunk_base, 16, 0, AUX);
}
#ifdef NVH_SLAB_NT
constexpr int kSlabAux = 2, kSlabAuxOdd = 2;
#elif defined(NVH_SLAB_NT_ODD)
constexpr int kSlabAux = 0, kSlabAuxOdd = 2;  // the odd launch's own slabs (the even launch's are touched into L2 by the odd one)
#else
constexpr int kSlabAux = 0, kSlabAuxOdd = 0;
#endif
constexpr int kStageAux = 2;  // the neighbours' quarters are read once, by this workgroup: 22.0 -> 20.8 us per pass (three streams)

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// Component `comp` of entry e of a book with an explicit table (Codebook.cs:262-281: lookup type 2, or type 1 with sequence_p -- the
// host built the table the way the reference does, host_setup.cpp): a slab record names such a book by lat_values = 0 and points at
// ONE word of the lattice pool, the table's offset in the VQ pool (host_slab.cpp: build_book_directory).  Entry form only (the digit
// form is for setups whose residue books are all lattice books).  The component comes from global memory (L2): these are the
// streams that took the descriptor kernels through round 5, which gather the same way.
__device__ __forceinline__ float table_value(const float* __restrict__ vq, const uint32_t* lat, unsigned dims, unsigned e, unsigned comp) {
  const unsigned es = e != NVH_ENTRY_SKIP ? e : 0u;  // ("no vector was added here": the caller adds +0.0f; the address stays inside the table)
  return vq[lat[0] + es * dims + comp];
}

// Residue adds of one frame (Residue1.cs:8-26, Residue2.cs:23-47 WriteVectors for lattice books).  A lane owns G consecutive
// vector components of one partition / channel (a "chain": the writes to it through the cascade stages are consecutive
// records) and keeps their running sums in registers from the first stage to the last: the reference's additions, in the
// reference's order per element, one LDS store per element.  Component i of the partition lies in entry i / dim of the
// stage's vector list as component i % dim; the entry's components are the base-lat_values digits of the entry number,
// peeled two at a time with exact reciprocal multiplies (consecutive pairs of one entry continue from the previous quotient).
struct FloorRef {  // the frame's floor curves as they lie in the slab (both channels), for the fused multiply
  const uint4* seg[2];
  const uint8_t* tab[2];
  int md[2];
  const float* s_db;
};

// One line segment of a rendered Floor1 curve as the slab holds it: x = x0 | xend << 16, y = the curve at x0, (w:z) = the
// curve's step per bin as a signed 32.32 fixed-point number.  Floor1.cs:316-340 draws y(x0 + k) = y0 + k b + sy floor(k r / adx)
// (b = dy / adx truncated, r = |dy| mod adx, sy = sign dy) with an error-term recurrence; the slab stores
// F = |b| 2^32 + ceil(2^32 r / adx), negated for a falling line, and the walk keeps (y : fraction) in one 64-bit register that
// it adds the step to: the carry out of the fraction is the recurrence's "err >= adx".  Exact: the fraction overestimates
// k r / adx by less than k 2^-32 <= 2^-19, and k r / adx is either an integer or at least 1 / adx >= 2^-13 below the next one.
// A falling line starts its fraction at 2^32 - 1, so that the borrow comes exactly where the rising line's carry would.
template <int NB>
__device__ __forceinline__ void floor_walk_fx(const uint4* __restrict__ seg, const uint8_t* __restrict__ segtab,
                                              const float* __restrict__ s_db, int x0, float m[NB]) {
  int sg = segtab[x0 >> 2];  // the last segment that starts at or before x0 (the slab writers)
  uint4 s = seg[sg];
  unsigned long long step = ((unsigned long long)s.w << 32) | s.z;
  const unsigned t = (unsigned)x0 - (s.x & 0xFFFFu);
  unsigned long long st = (((unsigned long long)s.y << 32) | (unsigned)((int)s.w >> 31)) + step * t;
  int xend = (int)(s.x >> 16);
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    // the next segment starts exactly here; the last segment ends at or beyond n/2, so x < n/2 never runs off the list
    if (x0 + q >= xend) {
      s = seg[++sg];
      step = ((unsigned long long)s.w << 32) | s.z;
      st = ((unsigned long long)s.y << 32) | (unsigned)((int)s.w >> 31);
      xend = (int)(s.x >> 16);
    }
    const int y = (int)(st >> 32);
    m[q] = s_db[y < 0 ? 0 : (y > 255 ? 255 : y)];  // out-of-range values were reported by floor_prepare (quirk B-7)
    st += step;
  }
}

// FUSE: the floor multiply (Floor1.cs:196-222) happens here, on the registers that hold a chain's finished sums, and only
// for bins some chain covers: every other bin of the cleared spectrum stays +0.0f, which is what 0 * curve gives anyway.
// RCH >= 3: Residue2 over RCH channels, G = 2 RCH: a lane owns two bins of every channel (component k = channel k % RCH of
// bin k / RCH, Residue2.cs:25-45).  RCH == 0: the two-channel interleave or a per-channel residue, told apart at run time.
// DIG: the entry section holds digit bytes (nvh_format.h: NVH_SLAB_RGEOM_DIGITS) -- a lane's G components of a cascade stage are G
// consecutive bytes of the record's run, each the byte offset of its float from the book's first word in the value pool: the
// stage is one LDS read of the bytes, then a read and an add per component.  Not DIG: uint16 entry numbers, digits peeled here
// (the form the GPU parser still writes).
template <int G, bool FUSE = false, int RCH = 0, int NT = SP_THREADS, bool DIG = false>
__device__ __forceinline__ void residue_walk(const float* slab, unsigned off_heads, unsigned off_rec, unsigned off_ent,
                                             const uint32_t* __restrict__ s_lat, float* spec, int half, unsigned nheads,
                                             unsigned lpc, unsigned lpc_magic, bool interleaved, unsigned flags, int tid,
                                             const FloorRef* F = nullptr, const float* __restrict__ vq = nullptr) {
  static_assert(RCH == 0 || G == 2 * RCH, "two bins of every channel per lane");
  const uint32_t* heads = reinterpret_cast<const uint32_t*>(slab + off_heads * 4);
  const uint2* recs = reinterpret_cast<const uint2*>(slab + off_rec * 4);
  const uint16_t* ent = reinterpret_cast<const uint16_t*>(slab + off_ent * 4);
  const bool sweep_couples = (flags & NVH_SLAB_SWEEP_COUPLES) != 0;
  const bool mg1 = (flags & NVH_SLAB_MG1) != 0;
  const unsigned total = nheads * lpc;
  for (unsigned idx = tid; idx < total; idx += NT) {
    const unsigned oq = lpc > 1 ? __umulhi(idx, lpc_magic) : idx;
    const unsigned g = idx - oq * lpc, i0 = g * G;  // first component of this lane's group inside the partition
    const unsigned hd = heads[oq];
    unsigned o = hd & 0xFFFFu;
    uint2 rec = recs[o];
    const unsigned xbase = hd >> 16;
    float a[G];
#pragma unroll
    for (int k = 0; k < G; ++k) a[k] = 0.0f;  // the spectrum was cleared and every bin belongs to one chain
    for (;;) {
      if constexpr (DIG) {
        // the book's values (+0.0f behind them: the slot of a vector that was never added, the identity on these sums)
        const char* vb = reinterpret_cast<const char*>(s_lat) + ((rec.y & 0xFFFu) << 2);
        const uint8_t* db = reinterpret_cast<const uint8_t*>(ent) + ((rec.x & 0xFFFFu) << 1) + i0;
        float v[G];
        if constexpr (G == 8) {
          const uint2 w = *reinterpret_cast<const uint2*>(db);  // runs are whole groups of eight: 8-byte aligned
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[k] = *reinterpret_cast<const float*>(vb + ((w.x >> (8 * k)) & 0xFFu));
            v[4 + k] = *reinterpret_cast<const float*>(vb + ((w.y >> (8 * k)) & 0xFFu));
          }
        } else {
#pragma unroll
          for (int k = 0; k < G; k += 2) {
            const unsigned h = *reinterpret_cast<const uint16_t*>(db + k);
            v[k] = *reinterpret_cast<const float*>(vb + (h & 0xFFu));
            v[k + 1] = *reinterpret_cast<const float*>(vb + (h >> 8));
          }
        }
#pragma unroll
        for (int k = 0; k < G; ++k) a[k] = a[k] + v[k];
        if (!(rec.y & 0x80000000u)) break;
        rec = recs[++o];
        continue;
      }
      const unsigned dims = (rec.y >> 20) & 31u, lv = (rec.y >> 12) & 0xFFu, dm16 = rec.x >> 16;
      const uint32_t* lat = s_lat + (rec.y & 0xFFFu);
      const uint16_t* eb = ent + (rec.x & 0xFFFFu);
      if (lv == 0u) {  // a book with an explicit table: the components out of the VQ pool (table_value)
#pragma unroll
        for (int k = 0; k < G; ++k) {
          const unsigned i = i0 + k;
          const unsigned j = (i * dm16) >> 16;  // i / dims
          const unsigned e = eb[j];
          const float v = table_value(vq, lat, dims, e, i - j * dims);
          a[k] = a[k] + (e != NVH_ENTRY_SKIP ? v : 0.0f);
        }
        if (!(rec.y & 0x80000000u)) break;
        rec = recs[++o];
        continue;
      }
      const unsigned lvm = lat[lv + 1];  // ceil(2^32 / lv): the second of the book's power magics (dims >= 2)
      // Branch-free on purpose: every entry of the group is fetched first, then the digits, then the lattice values, so
      // that the LDS round trips of the group overlap instead of queueing behind exec-mask regions.  A skipped entry
      // ("no vector was added here", quirks B-14 / B-16) adds +0.0f, which is the identity on these sums: they start at
      // +0.0f and a sum is -0.0f only when both operands are.
      unsigned e[G / 2], comp[G / 2];
#pragma unroll
      for (int k = 0; k < G; k += 2) {
        const unsigned i = i0 + k;
        const unsigned j = (i * dm16) >> 16;  // i / dims (i < 4096, dims <= 16: exact)
        comp[k / 2] = i - j * dims;
        e[k / 2] = eb[j];
      }
      // e / lv^comp where the group starts inside an entry (dims > G): one more lookup, for the first pair only
      const unsigned pw = lat[lv + comp[0]];
      unsigned q2 = 0;
      unsigned d[G];
#pragma unroll
      for (int k = 0; k < G; k += 2) {
        unsigned q;
        if (k == 0) q = comp[0] ? __umulhi(e[0], pw) : e[0];
        else q = comp[k / 2] ? q2 : e[k / 2];  // the same entry continues from the previous quotient, or the next one begins
        // two base-lv digits (lv == 1: the magic is 0 and so are q and both digits)
        const unsigned q1 = __umulhi(q, lvm);
        d[k] = q - __umul24(q1, lv);
        q2 = __umulhi(q1, lvm);
        d[k + 1] = q1 - __umul24(q2, lv);
      }
      float v[G];
#pragma unroll
      for (int k = 0; k < G; ++k) v[k] = __uint_as_float(lat[d[k]]);
#pragma unroll
      for (int k = 0; k < G; ++k) a[k] = a[k] + (e[k / 2] != NVH_ENTRY_SKIP ? v[k] : 0.0f);
      if (!(rec.y & 0x80000000u)) break;
      rec = recs[++o];
    }
    if constexpr (RCH >= 3) {
      // a[c] / a[RCH + c] = bins xb / xb + 1 of channel c
      const unsigned xb = xbase + 2 * g;
#pragma unroll
      for (int c = 0; c < RCH; ++c) {
        float* p = spec + (unsigned)c * (unsigned)half + xb;
        if (xb < (unsigned)half) p[0] = a[c];
        if (xb + 1 < (unsigned)half) p[1] = a[RCH + c];
      }
    } else
    if (interleaved) {
      // a[2m] / a[2m + 1] = bin xb + m of channel 0 / 1
      const unsigned xb = xbase + (i0 >> 1);
      if (sweep_couples) {  // Mapping.cs:137-182 on the pair a lane holds anyway (one uniform branch, not one per bin)
        if (!mg1) {
#pragma unroll
          for (int m = 0; m < G / 2; ++m) couple1(a[2 * m], a[2 * m + 1]);
        } else {
#pragma unroll
          for (int m = 0; m < G / 2; ++m) couple1(a[2 * m + 1], a[2 * m]);
        }
      }
      float* p0 = spec + xb;
      float* p1 = spec + (unsigned)half + xb;
      if constexpr (FUSE && G == 8) {  // xb is a multiple of 4 (the slab writers checked the residue's geometry)
        if (xb + 4 <= (unsigned)half) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            if (F->md[c] == 1) {
              float m[4];
              floor_walk_fx<4>(F->seg[c], F->tab[c], F->s_db, (int)xb, m);
#pragma unroll
              for (int q = 0; q < 4; ++q) a[2 * q + c] = a[2 * q + c] * m[q];
            } else if (F->md[c] == 2) {
#pragma unroll
              for (int q = 0; q < 4; ++q) a[2 * q + c] = 0.0f;  // Floor1.cs:218-221
            }
          }
          *reinterpret_cast<float4*>(p0) = make_float4(a[0], a[2], a[4 % G], a[6 % G]);
          *reinterpret_cast<float4*>(p1) = make_float4(a[1], a[3 % G], a[5 % G], a[7 % G]);
        }
      } else
      if (G == 8 && (xb & 3u) == 0 && xb + 4 <= (unsigned)half) {
        *reinterpret_cast<float4*>(p0) = make_float4(a[0], a[2], a[4 % G], a[6 % G]);
        *reinterpret_cast<float4*>(p1) = make_float4(a[1], a[3 % G], a[5 % G], a[7 % G]);
      } else {
#pragma unroll
        for (int m = 0; m < G / 2; ++m)
          if (xb + m < (unsigned)half) {
            p0[m] = a[2 * m];
            p1[m] = a[2 * m + 1];
          }
      }
    } else {
      const unsigned c = (rec.y >> 25) & 7u;
      const unsigned xb = xbase + i0;
      float* p = spec + c * (unsigned)half + xb;
      if constexpr (FUSE && G == 8) {  // xb is a multiple of 4
        if (xb + 8 <= (unsigned)half) {
          const bool c1 = c != 0;
          const int md = c1 ? F->md[1] : F->md[0];
          if (md == 1) {
            float m[8];
            floor_walk_fx<8>(c1 ? F->seg[1] : F->seg[0], c1 ? F->tab[1] : F->tab[0], F->s_db, (int)xb, m);
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q % G] = a[q % G] * m[q];
          } else if (md == 2) {
#pragma unroll
            for (int q = 0; q < G; ++q) a[q] = 0.0f;
          }
          *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2 % G], a[3 % G]);
          *reinterpret_cast<float4*>(p + 4) = make_float4(a[4 % G], a[5 % G], a[6 % G], a[7 % G]);
        } else if (xb + 4 <= (unsigned)half) {  // half = 4 (mod 8) cannot happen (powers of two >= 128): kept for clarity
          __builtin_trap();
        }
      } else
      if (G == 8 && (xb & 3u) == 0 && xb + 8 <= (unsigned)half) {
        *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2 % G], a[3 % G]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(a[4 % G], a[5 % G], a[6 % G], a[7 % G]);
      } else {
#pragma unroll
        for (int k = 0; k < G; ++k)
          if (xb + k < (unsigned)half) p[k] = a[k];
      }
    }
  }
}


// The same walk for the TWO frames of a frame group at once (digit form, eight components per lane, stereo Residue2 with the floor
// multiply fused: the shape of every libvorbis stereo stream): a lane takes its chain group of frame 0 AND its chain group of frame 1
// through the cascade stages in one loop, so that the two chains' LDS round trips -- record, digit bytes, eight values, per stage --
// are in flight together instead of one chain after the other (a workgroup's two walks were 10 k of its 27-31 k cycles, each a
// chain of dependent reads: profiles/r06_group_phases.txt).  Branch-free inside the loop: a chain that has ended (or a lane without
// a group in that frame) re-reads its last record and adds +0.0f, the identity on these sums.  Same additions, same order per
// element as residue_walk.
struct WalkFrame {
  const float* slab;
  unsigned off_heads, off_rec, off_ent, nheads, lpc, lpc_magic, flags;
  float* spec;
  int half;
  FloorRef F;
};

template <int NT>
__device__ __forceinline__ void residue_walk_two(const WalkFrame (&W)[2], const uint32_t* __restrict__ s_lat, int tid) {
  constexpr int G = 8;
  const uint32_t* heads[2];
  const uint2* recs[2];
  const uint8_t* ent[2];
  unsigned total[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    heads[f] = reinterpret_cast<const uint32_t*>(W[f].slab + W[f].off_heads * 4);
    recs[f] = reinterpret_cast<const uint2*>(W[f].slab + W[f].off_rec * 4);
    ent[f] = reinterpret_cast<const uint8_t*>(W[f].slab + W[f].off_ent * 4);
    total[f] = W[f].nheads * W[f].lpc;  // >= 1 (the caller)
  }
  const unsigned tmax = total[0] > total[1] ? total[0] : total[1];
  for (unsigned idx = tid; idx < tmax; idx += NT) {
    bool act[2], more[2];
    unsigned o[2], xbase[2], i0[2];
    uint2 rec[2];
    float a[2][G];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      act[f] = idx < total[f];
      const unsigned ii = act[f] ? idx : 0u;  // (a lane without a group in this frame walks group 0 and stores nothing)
      const unsigned oq = W[f].lpc > 1 ? __umulhi(ii, W[f].lpc_magic) : ii;
      i0[f] = (ii - oq * W[f].lpc) * G;
      const unsigned hd = heads[f][oq];
      o[f] = hd & 0xFFFFu;
      xbase[f] = hd >> 16;
      rec[f] = recs[f][o[f]];
      more[f] = act[f];
#pragma unroll
      for (int k = 0; k < G; ++k) a[f][k] = 0.0f;
    }
#ifdef NVH_ABL_NO_CHAIN2
    more[0] = more[1] = false;  // (ablation build: the walk without its cascade stages)
#endif
    while (more[0] | more[1]) {
      float v[2][G];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const char* vb = reinterpret_cast<const char*>(s_lat) + ((rec[f].y & 0xFFFu) << 2);
        const uint8_t* db = ent[f] + ((rec[f].x & 0xFFFFu) << 1) + i0[f];
        const uint2 w = *reinterpret_cast<const uint2*>(db);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[f][k] = *reinterpret_cast<const float*>(vb + ((w.x >> (8 * k)) & 0xFFu));
          v[f][4 + k] = *reinterpret_cast<const float*>(vb + ((w.y >> (8 * k)) & 0xFFu));
        }
      }
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        // (the +0.0f of a chain that has ended as a bit mask on the value, not as a select: the compiler turns a select into a
        // branch around that frame's loads, and the two frames' reads then queue up behind one another again)
        const uint32_t keep = more[f] ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int k = 0; k < G; ++k) a[f][k] = a[f][k] + __uint_as_float(__float_as_uint(v[f][k]) & keep);
        more[f] = more[f] && (rec[f].y & 0x80000000u) != 0u;
        o[f] += more[f] ? 1u : 0u;
        rec[f] = recs[f][o[f]];
      }
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (!act[f]) continue;
      // a[2m] / a[2m + 1] = bin xb + m of channel 0 / 1 (residue_walk: interleaved, FUSE, G == 8)
      const unsigned xb = xbase[f] + (i0[f] >> 1);
      if (W[f].flags & NVH_SLAB_SWEEP_COUPLES) {  // Mapping.cs:137-182
        if (!(W[f].flags & NVH_SLAB_MG1)) {
#pragma unroll
          for (int m = 0; m < G / 2; ++m) couple1(a[f][2 * m], a[f][2 * m + 1]);
        } else {
#pragma unroll
          for (int m = 0; m < G / 2; ++m) couple1(a[f][2 * m + 1], a[f][2 * m]);
        }
      }
      if (xb + 4 <= (unsigned)W[f].half) {
        // (the four curves of a lane -- two frames x two channels -- side by side, their table, segment and inverse_dB_table reads in
        // flight together, the segment behind each fetched with it: built and measured, round 6 -- 80 instead of 69 VGPRs, 220 -> 210 M
        // frames/s over three streams; these four chains one after the other stay)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#ifdef NVH_ABL_NO_FLOOR2
          if (false) {  // (ablation build: the walk without its floor multiply)
#else
          if (W[f].F.md[c] == 1) {
#endif
            float m[4];
            floor_walk_fx<4>(W[f].F.seg[c], W[f].F.tab[c], W[f].F.s_db, (int)xb, m);
#pragma unroll
            for (int q = 0; q < 4; ++q) a[f][2 * q + c] = a[f][2 * q + c] * m[q];
          } else if (W[f].F.md[c] == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[f][2 * q + c] = 0.0f;  // Floor1.cs:218-221
          }
        }
        *reinterpret_cast<float4*>(W[f].spec + xb) = make_float4(a[f][0], a[f][2], a[f][4], a[f][6]);
        *reinterpret_cast<float4*>(W[f].spec + (unsigned)W[f].half + xb) = make_float4(a[f][1], a[f][3], a[f][5], a[f][7]);
      }
    }
  }
}

// Residue2 whose partitions share bins (quirk B-1: `offset /= channels` truncates and chPtr restarts at 0 for every partition,
// Residue2.cs:25-27): component q of partition p lands in channel q % rch, bin (begin + p psz) / rch + q / rch, so
// when psz is not a multiple of rch the last bin of a partition is the first bin of the next one, and an element receives the
// vectors of both -- in the reference's order: stage by stage, and inside a stage the lower partition first (Residue0.cs:132-175).
// A lane owns one BIN (all rch channels of it) and merges the chains of the (at most two) partitions that touch it by cascade
// stage; chains lie in partition order, pchain[p] = chain of partition p (0xFFFF: none).  Slab section `bins`:
// uint32 begin, psz, nparts, bins a partition touches | uint16 pchain[nparts].
template <int NT, bool DIG = false>
__device__ __forceinline__ void residue_walk_bins(const float* slab, unsigned off_heads, unsigned off_rec, unsigned off_ent,
                                                  unsigned off_bins, unsigned psz_magic, const uint32_t* __restrict__ s_lat,
                                                  float* spec, int half, unsigned rch, int tid, const float* __restrict__ vq = nullptr) {
  const uint32_t* heads = reinterpret_cast<const uint32_t*>(slab + off_heads * 4);
  const uint2* recs = reinterpret_cast<const uint2*>(slab + off_rec * 4);
  const uint16_t* ent = reinterpret_cast<const uint16_t*>(slab + off_ent * 4);
  const uint32_t* prm = reinterpret_cast<const uint32_t*>(slab + off_bins * 4);
  const unsigned rbegin = __builtin_amdgcn_readfirstlane(prm[0]), psz = __builtin_amdgcn_readfirstlane(prm[1]);
  const unsigned nparts = __builtin_amdgcn_readfirstlane(prm[2]), cover = __builtin_amdgcn_readfirstlane(prm[3]);
  const uint16_t* pchain = reinterpret_cast<const uint16_t*>(prm + 4);
  if (nparts == 0) return;
  const unsigned rch_magic = (unsigned)((0x100000000ull + rch - 1) / rch);  // uniform, once per wavefront
  auto base_of = [&](unsigned p) { return __umulhi(rbegin + p * psz, rch_magic); };  // (begin + p psz) / rch: below 2^16 * 8
  const unsigned xfirst = base_of(0), xend = base_of(nparts - 1) + cover;
  for (unsigned X = xfirst + (unsigned)tid; X < xend; X += NT) {
    // the largest p whose first bin is at or before X
    unsigned phi = __umulhi((X + 1) * rch - 1 - rbegin, psz_magic);
    if (phi >= nparts) phi = nparts - 1;
    unsigned cb = pchain[phi], ca = 0xFFFFu;
    if (X >= base_of(phi) + cover) cb = 0xFFFFu;  // behind the last partition's bins
    if (phi >= 1 && X < base_of(phi - 1) + cover) ca = pchain[phi - 1];
    unsigned oa = 0, ob = 0, xa = 0, xbb = 0;
    uint2 ra = make_uint2(0u, 0u), rb = ra;
    bool va = ca != 0xFFFFu, vb = cb != 0xFFFFu;
    if (va) { const unsigned hd = heads[ca]; oa = hd & 0xFFFFu; xa = hd >> 16; ra = recs[oa]; }
    if (vb) { const unsigned hd = heads[cb]; ob = hd & 0xFFFFu; xbb = hd >> 16; rb = recs[ob]; }
    float a[NVH_SLAB_MAX_CH];
#pragma unroll
    for (int c = 0; c < NVH_SLAB_MAX_CH; ++c) a[c] = 0.0f;
    while (va || vb) {
      // the next vector write in the reference's order: the lower cascade stage, the lower partition (chain A) on a tie
      const bool take_a = va && (!vb || ((ra.y >> 28) & 7u) <= ((rb.y >> 28) & 7u));
      const uint2 rec = take_a ? ra : rb;
      const unsigned q0 = (X - (take_a ? xa : xbb)) * rch;  // first component of this bin inside the partition
      if constexpr (DIG) {
        // component q of the partition is byte q of the record's run (nvh_format.h: NVH_SLAB_RGEOM_DIGITS)
        const char* vb = reinterpret_cast<const char*>(s_lat) + ((rec.y & 0xFFFu) << 2);
        const uint8_t* db = reinterpret_cast<const uint8_t*>(ent) + ((rec.x & 0xFFFFu) << 1);
#pragma unroll
        for (int c = 0; c < NVH_SLAB_MAX_CH; ++c) {
          if ((unsigned)c < rch) {  // uniform
            const unsigned q = q0 + (unsigned)c;
            const bool in = q < psz;  // a partition's last bin may hold fewer than rch components
            const float v = *reinterpret_cast<const float*>(vb + db[in ? q : 0u]);
            a[c] = a[c] + (in ? v : 0.0f);
          }
        }
      } else {
      const unsigned dims = (rec.y >> 20) & 31u, lv = (rec.y >> 12) & 0xFFu, dm16 = rec.x >> 16;
      const uint32_t* lat = s_lat + (rec.y & 0xFFFu);
      const uint16_t* eb = ent + (rec.x & 0xFFFFu);
      const unsigned lvm = lat[lv + 1];
#pragma unroll
      for (int c = 0; c < NVH_SLAB_MAX_CH; ++c) {
        if ((unsigned)c < rch) {  // uniform
          const unsigned q = q0 + (unsigned)c;
          const bool in = q < psz;  // a partition's last bin may hold fewer than rch components
          const unsigned qq = in ? q : 0u;
          const unsigned j = (qq * dm16) >> 16, comp = qq - j * dims;
          const unsigned e = eb[j];
          float v;
          if (lv == 0u) {
            v = table_value(vq, lat, dims, e, comp);
          } else {
            const unsigned pw = lat[lv + comp];
            const unsigned qv = comp ? __umulhi(e, pw) : e;
            const unsigned dgt = qv - __umul24(__umulhi(qv, lvm), lv);
            v = __uint_as_float(lat[dgt]);
          }
          a[c] = a[c] + ((in && e != NVH_ENTRY_SKIP) ? v : 0.0f);  // +0.0f is the identity on these sums (they start at +0.0f)
        }
      }
      }
      if (take_a) {
        va = (ra.y & 0x80000000u) != 0;
        if (va) ra = recs[++oa];
      } else {
        vb = (rb.y & 0x80000000u) != 0;
        if (vb) rb = recs[++ob];
      }
    }
    if (X < (unsigned)half) {
#pragma unroll
      for (int c = 0; c < NVH_SLAB_MAX_CH; ++c)
        if ((unsigned)c < rch) spec[(unsigned)c * (unsigned)half + X] = a[c];
    }
  }
}

// The general bin walk (NvhSlabHdr::group == 1): everything the two walks above leave out -- Residue0 (entry j of a partition adds
// component d at offset j + d steps, Residue0.cs:180-201), books of odd dimension, Residue2 over one or two channels with
// partitions off the bin grid, several residue passes per frame (one per submap, each over every channel: Mapping.cs:122-134).
// The slab carries a group list (host_slab.cpp: residue_general): per (pass, channel) of a per-channel residue, or per pass of
// a Residue2, the residue's geometry and pchain[p] = the chain of partition p.  A lane owns one bin of one group and adds the
// vectors that land there in the reference's order (stage by stage, the lower partition first where two share a bin); a
// later pass continues from the sums the earlier one stored, behind a barrier.
template <int NT, int MAXC, bool DIG = false>
__device__ __forceinline__ void residue_walk_general(const float* slab, unsigned off_heads, unsigned off_rec, unsigned off_ent,
                                                     unsigned off_gen, const uint32_t* __restrict__ s_lat, float* spec, int half,
                                                     int tid, const float* __restrict__ vq = nullptr) {
  const uint32_t* heads = reinterpret_cast<const uint32_t*>(slab + off_heads * 4);
  const uint2* recs = reinterpret_cast<const uint2*>(slab + off_rec * 4);
  const uint16_t* ent = reinterpret_cast<const uint16_t*>(slab + off_ent * 4);
  const uint32_t* prm = reinterpret_cast<const uint32_t*>(slab + off_gen * 4);
  const unsigned ngroups = __builtin_amdgcn_readfirstlane(prm[0]);
  const uint16_t* pchain_all = reinterpret_cast<const uint16_t*>(prm + 4 + 8 * ngroups);
  unsigned cur_pass = 0;
  for (unsigned gi = 0; gi < ngroups; ++gi) {
    const uint32_t* G = prm + 4 + 8 * gi;
    const unsigned rbegin = __builtin_amdgcn_readfirstlane(G[0]), psz = __builtin_amdgcn_readfirstlane(G[1]);
    const unsigned nparts = __builtin_amdgcn_readfirstlane(G[2]), cover = __builtin_amdgcn_readfirstlane(G[3]);
    const unsigned geom = __builtin_amdgcn_readfirstlane(G[4]), psz_magic = __builtin_amdgcn_readfirstlane(G[5]);
    const uint16_t* pchain = pchain_all + __builtin_amdgcn_readfirstlane(G[6]);
    const unsigned rtype = geom & 15u, rch = (geom >> 4) & 15u, pass = (geom >> 8) & 15u, chan0 = (geom >> 12) & 15u;
    if (pass != cur_pass) {  // (uniform) the sums of the pass before are in the spectra
      __syncthreads();
      cur_pass = pass;
    }
    if (nparts == 0) continue;
    const unsigned rch_magic = rch > 1 ? (unsigned)((0x100000000ull + rch - 1) / rch) : 0u;
    auto base_of = [&](unsigned p) { const unsigned o = rbegin + p * psz; return rch > 1 ? __umulhi(o, rch_magic) : o; };
    const unsigned xfirst = base_of(0), xend = base_of(nparts - 1) + cover;
    for (unsigned X = xfirst + (unsigned)tid; X < xend; X += NT) {
      unsigned phi = __umulhi((X + 1) * rch - 1 - rbegin, psz_magic);  // the largest p whose first bin is at or before X
      if (phi >= nparts) phi = nparts - 1;
      unsigned cb = pchain[phi], ca = 0xFFFFu;
      if (X >= base_of(phi) + cover) cb = 0xFFFFu;
      if (phi >= 1 && X < base_of(phi - 1) + cover) ca = pchain[phi - 1];
      unsigned oa = 0, ob = 0, xa = 0, xbb = 0;
      uint2 ra = make_uint2(0u, 0u), rb = ra;
      bool va = ca != 0xFFFFu, vb = cb != 0xFFFFu;
      if (va) { const unsigned hd = heads[ca]; oa = hd & 0xFFFFu; xa = hd >> 16; ra = recs[oa]; }
      if (vb) { const unsigned hd = heads[cb]; ob = hd & 0xFFFFu; xbb = hd >> 16; rb = recs[ob]; }
      if (!va && !vb) continue;  // nothing lands here: the bin keeps what it holds
      float a[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        a[c] = 0.0f;
        if (pass != 0 && (unsigned)c < rch && X < (unsigned)half) a[c] = spec[(chan0 + (unsigned)c) * (unsigned)half + X];
      }
      while (va || vb) {
        const bool take_a = va && (!vb || ((ra.y >> 28) & 7u) <= ((rb.y >> 28) & 7u));
        const uint2 rec = take_a ? ra : rb;
        const unsigned q0 = (X - (take_a ? xa : xbb)) * rch;
        const unsigned dims = (rec.y >> 20) & 31u, lv = (rec.y >> 12) & 0xFFu, dm16 = rec.x >> 16;
        const uint32_t* lat = s_lat + (rec.y & 0xFFFu);
        const uint16_t* eb = ent + (rec.x & 0xFFFFu);
        const unsigned lvm = (!DIG && dims > 1) ? lat[lv + 1] : 0u;  // (a book of dimension 1 has no second power: the entry is the digit)
        // partition_size / dims (host_slab.cpp checked the reciprocal; dimension 1: the record's 16-bit field cannot hold 2^16)
        const unsigned steps = dims > 1 ? (psz * dm16) >> 16 : psz;
        // components this vector write covers: the partition, or -- Residue1 / Residue2, a dimension that does not divide it -- whole
        // entries, the last one running over into the next partition's elements (Residue1.cs:12-22, Residue2.cs:27-45)
        const unsigned span = (rtype == 0 || dims <= 1) ? psz : (((psz + dims - 1u) * dm16) >> 16) * dims;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          if ((unsigned)c < rch) {  // uniform
            const unsigned q = q0 + (unsigned)c;
            const bool in = q < span;
            const unsigned qq = in ? q : 0u;
            unsigned j, comp;
            if (rtype == 0) {  // component comp of entry j lies at j + comp * steps
              comp = __umulhi(qq * dims, psz_magic);
              j = qq - comp * steps;
            } else {
              j = dims > 1 ? (qq * dm16) >> 16 : qq;
              comp = qq - j * dims;
            }
            if constexpr (DIG) {  // byte j * dims + comp of the record's run (nvh_format.h: NVH_SLAB_RGEOM_DIGITS)
              const uint8_t* db = reinterpret_cast<const uint8_t*>(ent) + ((rec.x & 0xFFFFu) << 1);
              const unsigned pos = rtype == 0 ? j * dims + comp : qq;
              const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lat) + db[pos]);
              a[c] = a[c] + (in ? v : 0.0f);
            } else {
            const unsigned e = eb[j];
            float v;
            if (lv == 0u) {
              v = table_value(vq, lat, dims, e, comp);
            } else {
              const unsigned pw = lat[lv + comp];
              const unsigned qv = comp ? __umulhi(e, pw) : e;
              const unsigned dgt = qv - __umul24(__umulhi(qv, lvm), lv);
              v = __uint_as_float(lat[dgt]);
            }
            a[c] = a[c] + ((in && e != NVH_ENTRY_SKIP) ? v : 0.0f);
            }
          }
        }
        if (take_a) {
          va = (ra.y & 0x80000000u) != 0;
          if (va) ra = recs[++oa];
        } else {
          vb = (rb.y & 0x80000000u) != 0;
          if (vb) rb = recs[++ob];
        }
      }
      if (X < (unsigned)half) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if ((unsigned)c < rch) spec[(chan0 + (unsigned)c) * (unsigned)half + X] = a[c];
      }
    }
  }
}

}  // namespace

// NVH_EMIT_CARRY_OUT: the block that becomes the carried tail of the next batch (StreamDecoder's _prevPacketBuf), stored fully
// windowed like k_ola_compact stores it -- from the frame's own plane, which its wavefronts have just written: the caller's
// __syncthreads() is the workgroup-scope release / acquire between those stores and these loads (one CU, one L1).  One
// workgroup per batch.
template <int NT>
__device__ __forceinline__ void synth_carry_out(const NvhSynthArgs& A, const float* plane0, int n, int nch, unsigned exec_mask,
                                                unsigned window_off, int tid) {
  const float* __restrict__ w = A.windows + window_off;
  const int quads = n >> 2;
  for (int o = tid; o < quads * nch; o += NT) {
    const int c = o >= quads ? 1 : 0, g = o - c * quads;  // at most two channels
    *reinterpret_cast<float4*>(A.carry_out + (long long)c * A.block1 + 4 * g) =
        compact_value4(plane0 + (long long)c * A.block1, w, n, (int)((exec_mask >> c) & 1u), 4 * g);
  }
}

// NVH_EMIT_SELF_CARRY: the batch's first frame over the carried tail of the batch before, which was windowed when it was stored
// (k_ola_compact's prev_full case: no second window multiply, the tail in time order).  The frame's own first quarter A lies in
// its channel's dead transform slice (synth_emit).  One workgroup per batch: out of line, so that the steady-state loop keeps
// its registers.
template <int NT>
__device__ __forceinline__ void synth_self_carry(const NvhSynthArgs& A, const float* spec, int n, int nch, unsigned window_off,
                                              unsigned out_pos, int tid, int cstride = 0) {
  const int half = n >> 1;
  if (cstride == 0) cstride = half;  // floats between the channels' first quarters
  const float* __restrict__ w = A.windows + window_off;
  float* out = A.pcm + (long long)out_pos * nch;
  int clipped = 0;
  for (int g = tid; g < (n >> 4); g += NT) {
    const int i0 = 4 * g;
    const float4 wf = *reinterpret_cast<const float4*>(w + i0);
    const float4 wm = *reinterpret_cast<const float4*>(w + (half - 4 - i0));
    float fwd[8], mir[8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < nch) {
        const float4 a = *reinterpret_cast<const float4*>(spec + c * cstride + i0);
        const float* cp = A.carry + (long long)c * A.block1;
        const float4 tt = *reinterpret_cast<const float4*>(cp + half + i0);
        const float4 r = *reinterpret_cast<const float4*>(cp + (n - 4 - i0));
        float4 v = make_float4(a.x * wf.x, a.y * wf.y, a.z * wf.z, a.w * wf.w);
        v.x = v.x + tt.x; v.y = v.y + tt.y; v.z = v.z + tt.z; v.w = v.w + tt.w;
        float4 u = make_float4(-a.w * wm.x, -a.z * wm.y, -a.y * wm.z, -a.x * wm.w);
        u.x = u.x + r.x; u.y = u.y + r.y; u.z = u.z + r.z; u.w = u.w + r.w;
        if (A.clip) {
          v.x = clip_value(v.x, &clipped); v.y = clip_value(v.y, &clipped);
          v.z = clip_value(v.z, &clipped); v.w = clip_value(v.w, &clipped);
          u.x = clip_value(u.x, &clipped); u.y = clip_value(u.y, &clipped);
          u.z = clip_value(u.z, &clipped); u.w = clip_value(u.w, &clipped);
        }
        fwd[c] = v.x; fwd[2 + c] = v.y; fwd[4 + c] = v.z; fwd[6 + c] = v.w;
        mir[c] = u.x; mir[2 + c] = u.y; mir[4 + c] = u.z; mir[6 + c] = u.w;
      }
    }
    // (a 64 x 2 transposition through LDS in front of these stores, so that every instruction writes whole lines the way the wide
    // kernel's emission does, was tried: 202 -> 167 M frames/s -- nine spilled registers and four more wavefront syncs cost more than
    // half-line streaming stores do)
    if (nch == 2) {
      float4* of = reinterpret_cast<float4*>(out) + 2 * (long long)g;
      float4* om = reinterpret_cast<float4*>(out) + 2 * (long long)((n >> 3) - 1 - g);
      pcm_store4(of, fwd[0], fwd[1], fwd[2], fwd[3]);
      pcm_store4(of + 1, fwd[4], fwd[5], fwd[6], fwd[7]);
      pcm_store4(om, mir[0], mir[1], mir[2], mir[3]);
      pcm_store4(om + 1, mir[4], mir[5], mir[6], mir[7]);
    } else {
      pcm_store4(reinterpret_cast<float4*>(out) + g, fwd[0], fwd[2], fwd[4], fwd[6]);
      pcm_store4(reinterpret_cast<float4*>(out) + ((n >> 3) - 1 - g), mir[0], mir[2], mir[4], mir[6]);
    }
  }
  if (A.clip) report_clipped(clipped, A.clipped_flag);
}

// ---- paired emission (nvh_format.h: NVH_EMIT_*) ---------------------------------------------------------------------------
// An even frame of the second launch: inverse MDCT into registers, then the overlap-add, interleave and clip of the steady-state
// overlaps it takes part in -- Mode.cs:160-166 windows, StreamDecoder.cs:532-541 adds, StreamDecoder.cs:391-415 / Utils.cs:30-43
// interleave + clip, with ola_sym's arithmetic (kernels.hip: sample times i and n/2 - 1 - i of an overlap need exactly the first
// quarter A[i] of the later block and the third quarter B[i] of the earlier one, Mdct.cs:275-303):
//   SELF: PCM of this frame      = A(this) windowed + B(frame - 1) windowed
//   NEXT: PCM of the next frame  = A(frame + 1) windowed + B(this) windowed
// A(frame + 1) and B(frame - 1) lie in the work planes since the first launch; they come in by LDS-DMA over the constants and
// the slab -- dead once the chain walk is through -- while the transform runs; the frame's own quarters go from the transform's
// registers into the channel's dead transform slice, and every lane of the workgroup then overlap-adds, clips and interleaves
// one group of sample times of one overlap, straight into 16-byte vectors of PCM.
// The frame's own plane is written only when k_ola_compact still needs it (not both overlaps emitted here).
template <int NT>
__device__ __forceinline__ void synth_emit(const NvhSynthArgs& A, float* smem, float* spec, const uint32_t* s_chan, int n, int nch,
                                           unsigned frame, int sl, bool emit_self, bool emit_next, bool self_carry, bool carry_out,
                                           unsigned exec_mask, float* planes,
                                           const float* Aa, const float* Bb, const float* Cc, const float* TW, int tid,
                                           long long* stamps = nullptr) {
  const int wv = tid >> 6, lane = tid & 63, half = n >> 1;
  // profiling builds: shader-clock stamps of thread 0 (15 staging issued, 16 transform done, 17 behind the barrier, 18 emitted;
  // 10..14: the transform's own, imdct_wave.h)
#define EM_T(k) do { if (stamps && tid == 0) stamps[k] = clock64(); } while (0)
  // parameters of the overlaps, out of the slab before the staging overwrites it
  const unsigned w_self = __builtin_amdgcn_readfirstlane(s_chan[2]), wp_self = __builtin_amdgcn_readfirstlane(s_chan[3]);
  const unsigned w_next = __builtin_amdgcn_readfirstlane(s_chan[4]), wp_next = w_self;  // (the next frame's overlap source is this frame)
  const unsigned out_self = __builtin_amdgcn_readfirstlane(s_chan[6]), out_next = __builtin_amdgcn_readfirstlane(s_chan[7]);
  __syncthreads();  // the chain walk is through everywhere: constants and slab are dead, the spectra complete (the transform
                    // below is in place over the wavefront's own channel and needs no workgroup barrier of its own)
  // ---- stage the neighbours' quarters: per channel n/8 16-byte units, B(frame - 1) then A(frame + 1) ----
  float* stage = smem;
  {
    const int per_ch = n >> 3, q16 = n >> 4, sh = 28 - __clz(n);  // per_ch = 1 << sh
    const int units = nch * per_ch;
    for (int u0 = wv * 64; u0 < units; u0 += NT) {
      const int u = u0 + lane;
      if (u < units) {
        const int c = u >> sh, r = u & (per_ch - 1);
        const bool is_a = r >= q16;
#ifdef NVH_ABL_NO_STAGE
        if (false) {
#else
        if (is_a ? emit_next : (emit_self && !self_carry)) {
#endif
          const float* src = is_a ? A.work + ((long long)(frame + 1) * nch + c) * A.block1 + 4 * (r - q16)
                                  : A.work + ((long long)(frame - 1) * nch + c) * A.block1 + half + 4 * r;
          dma16<kStageAux>(reinterpret_cast<const uint4*>(src), stage + 4 * u0);
        }
      }
    }
  }
  EM_T(15);
  // ---- inverse MDCT of this wavefront's channel, the two independent quarters kept in registers ----
  float4 ca[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, cb[2] = {ca[0], ca[0]};
  const bool pon = lane < (n >> 5);  // lanes with output: four values of each quarter at 4 i8, i8 = lane and i8 = n/16 - 1 - lane
  const int i8a = lane, i8b = (n >> 4) - 1 - lane;
#ifdef NVH_ABL_NO_XFORM_ALL
  const bool xform = false;
#else
  const bool xform = wv < nch;  // every channel of an emitting frame executes (host: steady)
#endif
  if (xform) {
    const float* X = spec + wv * half;
    float* scratch = spec + wv * half - (nch - 1 - wv) * (n >> 4);
    auto sink = [&](int slot, int, float4 v) {  // slot = 4 h + q: q = 0 first quarter, q = 2 third quarter (1, 3: their mirrors)
      if (slot == 0) ca[0] = v; else if (slot == 2) cb[0] = v; else if (slot == 4) ca[1] = v; else if (slot == 6) cb[1] = v;
    };
    switch (n) {
      case 256: imdct_wave_sink<8, false, decltype(sink), true, false, false, kSynthPFEmit>(X, nullptr, scratch, Aa, Bb, Cc, TW, lane, sink, (stamps && wv == 0) ? stamps + 10 : nullptr, kAblSkip); break;
      case 512: imdct_wave_sink<9, false, decltype(sink), true, false, false, kSynthPFEmit>(X, nullptr, scratch, Aa, Bb, Cc, TW, lane, sink, (stamps && wv == 0) ? stamps + 10 : nullptr, kAblSkip); break;
      case 1024: imdct_wave_sink<10, false, decltype(sink), true, false, false, kSynthPFEmit>(X, nullptr, scratch, Aa, Bb, Cc, TW, lane, sink, (stamps && wv == 0) ? stamps + 10 : nullptr, kAblSkip); break;
      case 2048: imdct_wave_sink<11, false, decltype(sink), true, false, false, kSynthPFEmit>(X, nullptr, scratch, Aa, Bb, Cc, TW, lane, sink, (stamps && wv == 0) ? stamps + 10 : nullptr, kAblSkip); break;
      default: __builtin_trap();
    }
    if (pon) {
      if (!(emit_self && emit_next)) {  // k_ola_compact (or the carried tail) still reads this frame's plane
        float* plane = planes + (long long)wv * A.block1;
        *reinterpret_cast<float4*>(plane + 4 * i8a) = ca[0];
        *reinterpret_cast<float4*>(plane + 4 * i8b) = ca[1];
        *reinterpret_cast<float4*>(plane + half + 4 * i8a) = cb[0];
        *reinterpret_cast<float4*>(plane + half + 4 * i8b) = cb[1];
      }
      // the channel's own quarters, for the overlap-add below: A(this) in [0, n/4), B(this) in [n/4, n/2) of its dead slice
      float* own = spec + wv * half;
      *reinterpret_cast<float4*>(own + 4 * i8a) = ca[0];
      *reinterpret_cast<float4*>(own + 4 * i8b) = ca[1];
      *reinterpret_cast<float4*>(own + (half >> 1) + 4 * i8a) = cb[0];
      *reinterpret_cast<float4*>(own + (half >> 1) + 4 * i8b) = cb[1];
    }
  }
  EM_T(16);
  __syncthreads();  // drains the staging DMA (vmcnt(0) in front of the barrier): all four quarters of every channel are in LDS
  EM_T(17);
  if (carry_out) synth_carry_out<NT>(A, planes, n, nch, exec_mask, w_self, tid);  // (such a frame has no NEXT: its plane was written)
  if (self_carry) synth_self_carry<NT>(A, spec, n, nch, w_self, out_self, tid);    // the batch's first frame
  // ---- overlap-add + interleave + clip, every lane of the workgroup: lane task = (overlap, group of four compact indices i0);
  // it produces sample times i0 .. i0 + 3 and n/2 - 4 - i0 .. n/2 - 1 - i0 of every channel (kernels.hip: ola_sym) ----
  int clipped = 0;
  const int groups = n >> 4;  // per overlap
  // One task: NX = which overlap (a wave-uniform value when a wavefront's 64 tasks lie inside one overlap, i.e. n >= 1024: the
  // window / output / LDS bases are then scalar selects instead of per-lane 64-bit arithmetic).
  auto task = [&](const bool nx, const int g) {
    const int i0 = 4 * g;
    const float* __restrict__ w = A.windows + (nx ? w_next : w_self);
    const float* __restrict__ wp = A.windows + (nx ? wp_next : wp_self);
    // (fetching these in front of the barrier above, next to the staging DMA, was tried: the kernel sits at its 64-VGPR cap and
    // spills 32-48 registers for it)
#ifdef NVH_ABL_NO_WINDOW_LOAD
    const float4 wf = make_float4(0.5f, 0.25f, 0.125f, 0.75f), wm = wf, pf = wf, pm = wf;
    (void)w; (void)wp;
#else
    const float4 wf = *reinterpret_cast<const float4*>(w + i0);
    const float4 wm = *reinterpret_cast<const float4*>(w + (half - 4 - i0));
    const float4 pf = *reinterpret_cast<const float4*>(wp + (half + i0));
    const float4 pm = *reinterpret_cast<const float4*>(wp + (n - 4 - i0));
#endif
    float fwd[8], mir[8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < nch) {
        // SELF: a = A(this), b = B(frame - 1);  NEXT: a = A(frame + 1), b = B(this)
        const float* pa = nx ? stage + c * half + (half >> 1) : spec + c * half;
        const float* pb = nx ? spec + c * half + (half >> 1) : stage + c * half;
        const float4 a = *reinterpret_cast<const float4*>(pa + i0);
        const float4 b = *reinterpret_cast<const float4*>(pb + i0);
        float4 v = make_float4(a.x * wf.x, a.y * wf.y, a.z * wf.z, a.w * wf.w);
        const float4 tt = make_float4(b.x * pf.x, b.y * pf.y, b.z * pf.z, b.w * pf.w);
        v.x = v.x + tt.x; v.y = v.y + tt.y; v.z = v.z + tt.z; v.w = v.w + tt.w;
        float4 u = make_float4(-a.w * wm.x, -a.z * wm.y, -a.y * wm.z, -a.x * wm.w);
        const float4 r = make_float4(b.w * pm.x, b.z * pm.y, b.y * pm.z, b.x * pm.w);
        u.x = u.x + r.x; u.y = u.y + r.y; u.z = u.z + r.z; u.w = u.w + r.w;
#ifndef NVH_ABL_NO_CLIP
        // (a pre-test on the task's largest |x| in front of the compares and selects was tried twice: a wavefront's 1024 samples
        // of loud material nearly always hold one that clips, so the slow path runs anyway)
        if (A.clip) {
          v.x = clip_value(v.x, &clipped); v.y = clip_value(v.y, &clipped);
          v.z = clip_value(v.z, &clipped); v.w = clip_value(v.w, &clipped);
          u.x = clip_value(u.x, &clipped); u.y = clip_value(u.y, &clipped);
          u.z = clip_value(u.z, &clipped); u.w = clip_value(u.w, &clipped);
        }
#endif
        fwd[c] = v.x; fwd[2 + c] = v.y; fwd[4 + c] = v.z; fwd[6 + c] = v.w;
        mir[c] = u.x; mir[2 + c] = u.y; mir[4 + c] = u.z; mir[6 + c] = u.w;
      }
    }
    float* out = A.pcm + (long long)(nx ? out_next : out_self) * nch;
#ifdef NVH_ABL_PCM_SMALL
    out = A.pcm + (long long)((nx ? out_next : out_self) & 0x7FFF) * nch;  // (ablation build: every frame's PCM into the same 256 KB)
#endif
#ifdef NVH_ABL_NO_PCM_STORE
    if (fwd[0] != 1.2345e-30f) return;  // (ablation build: the arithmetic kept alive, the stores left out)
#endif
    if (nch == 2) {
      float4* of = reinterpret_cast<float4*>(out) + 2 * (long long)g;
      float4* om = reinterpret_cast<float4*>(out) + 2 * (long long)((n >> 3) - 1 - g);
      pcm_store4(of, fwd[0], fwd[1], fwd[2], fwd[3]);
      pcm_store4(of + 1, fwd[4], fwd[5], fwd[6], fwd[7]);
      pcm_store4(om, mir[0], mir[1], mir[2], mir[3]);
      pcm_store4(om + 1, mir[4], mir[5], mir[6], mir[7]);
    } else {
      pcm_store4(reinterpret_cast<float4*>(out) + g, fwd[0], fwd[2], fwd[4], fwd[6]);
      pcm_store4(reinterpret_cast<float4*>(out) + ((n >> 3) - 1 - g), mir[0], mir[2], mir[4], mir[6]);
    }
  };
  const bool do_self = emit_self && !self_carry;
#ifdef NVH_ABL_NO_EMIT_LOOP
  for (int t = tid; t < 0; t += NT) {
#else
  for (int t = tid; t < 2 * groups; t += NT) {
#endif
    if (groups >= 64) {  // (uniform) a wavefront's tasks belong to one overlap
      const bool nx = __builtin_amdgcn_readfirstlane((int)(t >= groups)) != 0;
      if (nx ? emit_next : do_self) task(nx, nx ? t - groups : t);
    } else {
      const bool nx = t >= groups;
      if (nx ? emit_next : do_self) task(nx, nx ? t - groups : t);
    }
  }
  if (A.clip) report_clipped(clipped, A.clipped_flag);
  EM_T(18);
#undef EM_T
}

// ---- paired emission for more than two channels (k_synth8_emit) ------------------------------------------------------------
// The even frames of a wide batch: a frame's compact planes are 48 KB for six channels at n = 4096, so neither the neighbours'
// quarters nor the frame's own fit the LDS next to the transforms' slices, and nothing stays in registers.  The workgroup stores
// its planes like every other, and then -- the slices are dead -- does what k_ola_compact would do for the two steady-state
// overlaps it takes part in (its own first half over frame f - 1's second half: SELF; frame f + 1's first half over its own
// second half: NEXT; Mode.cs:160-166 windows, StreamDecoder.cs:532-541 adds, :391-415 / Utils.cs:30-43 interleave + clip,
// kernels.hip: ola_sym_lds's arithmetic): a lane takes one (channel, group of four sample times), reads the later block's first
// quarter and the earlier block's third quarter -- its own from the L2 it has just written, the neighbour's from the odd launch --
// and puts its eight results into channel-planar LDS rows; behind a barrier the rows leave as 16-byte vectors of interleaved,
// clipped PCM.  The overlap-add's memory phase then runs inside the synthesis kernel, next to other workgroups' arithmetic,
// instead of as a launch of its own (k_ola_compact: 42 us per 2048 six-channel frames), and half the planes are read from L2.
template <int NT>
__device__ __forceinline__ void synth_emit8(const NvhSynthArgs& A, float* s_run, int n, int nch, unsigned frame, unsigned ef,
                                            unsigned exec_mask, int tid) {
  // One round per overlap: the rows of ALL groups fit the dead slices (2 x nch x n/4 floats: 48 KB for six channels at 4096),
  // so an overlap costs two barriers, and a lane's K tasks have their loads in flight together -- with one run of 64 groups
  // per round (k_ola_compact's shape) the eight rounds' round trips stood one behind the other: 87 -> 141 us for the pair of
  // launches on C4, more than the k_ola_compact launch they replace.
  constexpr int K = 3;
  const int half = n >> 1, groups = n >> 4, gsh = 31 - __clz(groups), RUN = n >> 2;  // RUN = 4 * groups
  const unsigned ch_magic = (unsigned)((0x100000000ull + (unsigned)nch - 1) / (unsigned)nch);
  const int total = groups * nch;
  int clipped = 0;
  for (int ov = 0; ov < 2; ++ov) {
    if (ov == 0 ? !(ef & NVH_EMIT_SELF) : !(ef & NVH_EMIT_NEXT)) continue;  // uniform
    const NvhFrame* fr = A.frames + frame + ov;  // the frame whose PCM this overlap is
    const bool from_carry = ov == 0 && (ef & NVH_EMIT_SELF_CARRY);  // the batch's first frame over the carried tail (stored windowed)
    const float* cur = A.work + (long long)(frame + ov) * nch * A.block1;  // the later block: its first quarter
    const float* prev = from_carry ? A.carry : A.work + (long long)(frame + ov - 1) * nch * A.block1;  // the earlier block
    const float* __restrict__ w = A.windows + fr->window_off;
    const float* __restrict__ wp = A.windows + fr->ov_window_off;
    float* out = A.pcm + fr->out_pos * nch;
    float* sF = s_run;
    float* sM = s_run + nch * RUN;
    for (int t0 = tid; t0 < total; t0 += K * NT) {
      float4 wf[K], wm[K], pf[K], pm[K], a[K], b[K], bm[K];
      int cc[K], gl[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int t = t0 + k * NT < total ? t0 + k * NT : total - 1;  // (a lane without a k-th task repeats the last one: same values)
        cc[k] = t >> gsh;
        gl[k] = t - (cc[k] << gsh);
        const int i0 = 4 * gl[k];
        wf[k] = *reinterpret_cast<const float4*>(w + i0);
        wm[k] = *reinterpret_cast<const float4*>(w + (half - 4 - i0));
        a[k] = *reinterpret_cast<const float4*>(cur + (long long)cc[k] * A.block1 + i0);
        if (from_carry) {  // time order, already windowed: samples half + i0 .. and n - 4 - i0 ..
          b[k] = *reinterpret_cast<const float4*>(prev + (long long)cc[k] * A.block1 + half + i0);
          bm[k] = *reinterpret_cast<const float4*>(prev + (long long)cc[k] * A.block1 + (n - 4 - i0));
        } else {
          pf[k] = *reinterpret_cast<const float4*>(wp + (half + i0));
          pm[k] = *reinterpret_cast<const float4*>(wp + (n - 4 - i0));
          b[k] = *reinterpret_cast<const float4*>(prev + (long long)cc[k] * A.block1 + half + i0);
        }
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float4 v = make_float4(a[k].x * wf[k].x, a[k].y * wf[k].y, a[k].z * wf[k].z, a[k].w * wf[k].w);
        float4 u = make_float4(-a[k].w * wm[k].x, -a[k].z * wm[k].y, -a[k].y * wm[k].z, -a[k].x * wm[k].w);
        if (from_carry) {
          v.x = v.x + b[k].x; v.y = v.y + b[k].y; v.z = v.z + b[k].z; v.w = v.w + b[k].w;
          u.x = u.x + bm[k].x; u.y = u.y + bm[k].y; u.z = u.z + bm[k].z; u.w = u.w + bm[k].w;
        } else {
          const float4 t = make_float4(b[k].x * pf[k].x, b[k].y * pf[k].y, b[k].z * pf[k].z, b[k].w * pf[k].w);
          v.x = v.x + t.x; v.y = v.y + t.y; v.z = v.z + t.z; v.w = v.w + t.w;
          const float4 r = make_float4(b[k].w * pm[k].x, b[k].z * pm[k].y, b[k].y * pm[k].z, b[k].x * pm[k].w);
          u.x = u.x + r.x; u.y = u.y + r.y; u.z = u.z + r.z; u.w = u.w + r.w;
        }
        *reinterpret_cast<float4*>(sF + cc[k] * RUN + 4 * gl[k]) = v;                 // sample times 4 gl ..
        *reinterpret_cast<float4*>(sM + cc[k] * RUN + 4 * (groups - 1 - gl[k])) = u;  // sample times n/2 - 4 - 4 gl ..
      }
    }
    __syncthreads();
    // the forward rows hold sample times [0, n/4), the mirrored rows [n/4, n/2): together the frame's n/2 samples in time order
    const int nvec = total;  // 16-byte vectors per half: n/4 sample times x nch channels / 4
    float4* oF = reinterpret_cast<float4*>(out);
    float4* oM = reinterpret_cast<float4*>(out + (long long)(half >> 1) * nch);
    for (int j = tid; j < 2 * nvec; j += NT) {
      const bool mir = j >= nvec;
      const int jj = mir ? j - nvec : j;
      const float* sr = mir ? sM : sF;
      float e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned idx = 4u * (unsigned)jj + (unsigned)k;  // position in the half's interleaved floats
        const unsigned tt = __umulhi(idx, ch_magic), c = idx - tt * (unsigned)nch;  // idx < 2^16, nch <= 8: exact
        float x = sr[c * (unsigned)RUN + tt];
        if (A.clip) x = clip_value(x, &clipped);
        e[k] = x;
      }
      pcm_store4((mir ? oM : oF) + jj, e[0], e[1], e[2], e[3]);
    }
    __syncthreads();
  }
  if (A.clip) report_clipped(clipped, A.clipped_flag);
}

// NVH_EMIT_CARRY_OUT for wide frames: the block that becomes the carried tail of the next batch, fully windowed, from the frame's
// own planes (the caller's barrier orders the plane stores in front of these loads).
template <int NT>
__device__ __forceinline__ void synth_carry_out8(const NvhSynthArgs& A, const float* plane0, int n, int nch, unsigned exec_mask,
                                                 unsigned window_off, int tid) {
  const float* __restrict__ w = A.windows + window_off;
  const int quads = n >> 2, qsh = 31 - __clz(quads);
  for (int o = tid; o < quads * nch; o += NT) {
    const int c = o >> qsh, g = o - (c << qsh);
    *reinterpret_cast<float4*>(A.carry_out + (long long)c * A.block1 + 4 * g) =
        compact_value4(plane0 + (long long)c * A.block1, w, n, (int)((exec_mask >> c) & 1u), 4 * g);
  }
}

// ---- paired emission for wide frames, direct form (round 5) ------------------------------------------------------------------
// A steady-state even frame of a wide batch (both of its overlaps emitted here, no carried tail on either side): the transforms
// keep their two independent quarters in registers until the output stage's gathers are through, then put them into the
// channel's own -- dead -- slice (A at [0, n/4), B at [n/4, n/2)); behind one barrier every lane of the workgroup takes one
// (overlap, group of four sample times) for ALL channels: own quarters from LDS, the neighbour's from the odd launch's planes,
// windows, Mode.cs:160-166 / StreamDecoder.cs:532-541 / :391-415 / Utils.cs:30-43 in ola_sym's arithmetic, and the 4 x CH
// interleaved samples of each half leave as CH 16-byte streaming stores.  The frame writes no plane, nothing is read back, no
// LDS rows, no second and third barrier (synth_emit8 keeps the frames with a carried tail or a single overlap).
template <int LD>
__device__ __forceinline__ void imdct_keep_quarters(const float* X, float* slice, const float* Aa, const float* Bb, const float* Cc,
                                                    const float* TW, int lane) {
  constexpr int n = 1 << LD, ITER = ((n >> 5) + 63) / 64;
  float4 ka[ITER][2], kb[ITER][2];
  auto sink = [&](int slot, int, float4 v) {
#pragma unroll
    for (int i = 0; i < ITER; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (slot == 8 * i + 4 * h) ka[i][h] = v;
        else if (slot == 8 * i + 4 * h + 2) kb[i][h] = v;
      }
  };
  imdct_wave_sink<LD, false, decltype(sink), true, true, false, true>(X, nullptr, slice, Aa, Bb, Cc, TW, lane, sink);
  wave_sync();  // the output stage's gathers are through: the slice is dead
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int p = lane + 64 * i;
    if (p < (n >> 5)) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i8 = h == 0 ? p : (n >> 4) - 1 - p;
        *reinterpret_cast<float4*>(slice + 4 * i8) = ka[i][h];
        *reinterpret_cast<float4*>(slice + (n >> 2) + 4 * i8) = kb[i][h];
      }
    }
  }
}

template <int NT, int CH>
__device__ __forceinline__ void synth_emit8_direct(const NvhSynthArgs& A, float* smem, int n, unsigned frame, int tid) {
  const int half = n >> 1, groups = n >> 4, slice = half + (n >> 4);
  const NvhFrame* fs = A.frames + frame;
  const unsigned w_self = fs[0].window_off, wp_self = fs[0].ov_window_off, w_next = fs[1].window_off, wp_next = fs[1].ov_window_off;
  const long long o_self = fs[0].out_pos, o_next = fs[1].out_pos;
  int clipped = 0;
  auto task = [&](const bool nx, const int g, const bool whole_wave) {
    const int i0 = 4 * g;
    const float* __restrict__ w = A.windows + (nx ? w_next : w_self);
    const float* __restrict__ wp = A.windows + (nx ? wp_next : wp_self);
    const float4 wf = *reinterpret_cast<const float4*>(w + i0);
    const float4 wm = *reinterpret_cast<const float4*>(w + (half - 4 - i0));
    const float4 pf = *reinterpret_cast<const float4*>(wp + (half + i0));
    const float4 pm = *reinterpret_cast<const float4*>(wp + (n - 4 - i0));
    // the neighbour's quarter of every channel first (one round trip for the lot): SELF: B(frame - 1), NEXT: A(frame + 1)
    const float* nb = nx ? A.work + (long long)(frame + 1) * CH * A.block1 + i0 : A.work + (long long)(frame - 1) * CH * A.block1 + half + i0;
    float4 q[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) q[c] = stream_load4(nb + (long long)c * A.block1);  // read once, by this lane
    float fwd[4 * CH], mir[4 * CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float* own = smem + c * slice;
      const float4 o = *reinterpret_cast<const float4*>(own + (nx ? (half >> 1) : 0) + i0);  // NEXT: B(this), SELF: A(this)
      const float4 a = nx ? q[c] : o, b = nx ? o : q[c];
      float4 v = make_float4(a.x * wf.x, a.y * wf.y, a.z * wf.z, a.w * wf.w);
      const float4 tt = make_float4(b.x * pf.x, b.y * pf.y, b.z * pf.z, b.w * pf.w);
      v.x = v.x + tt.x; v.y = v.y + tt.y; v.z = v.z + tt.z; v.w = v.w + tt.w;
      float4 u = make_float4(-a.w * wm.x, -a.z * wm.y, -a.y * wm.z, -a.x * wm.w);
      const float4 r = make_float4(b.w * pm.x, b.z * pm.y, b.y * pm.z, b.x * pm.w);
      u.x = u.x + r.x; u.y = u.y + r.y; u.z = u.z + r.z; u.w = u.w + r.w;
      if (A.clip) {
        v.x = clip_value(v.x, &clipped); v.y = clip_value(v.y, &clipped);
        v.z = clip_value(v.z, &clipped); v.w = clip_value(v.w, &clipped);
        u.x = clip_value(u.x, &clipped); u.y = clip_value(u.y, &clipped);
        u.z = clip_value(u.z, &clipped); u.w = clip_value(u.w, &clipped);
      }
      fwd[0 * CH + c] = v.x; fwd[1 * CH + c] = v.y; fwd[2 * CH + c] = v.z; fwd[3 * CH + c] = v.w;
      mir[0 * CH + c] = u.x; mir[1 * CH + c] = u.y; mir[2 * CH + c] = u.z; mir[3 * CH + c] = u.w;
    }
    float4* out = reinterpret_cast<float4*>(A.pcm + (nx ? o_next : o_self) * CH);
    float4* of = out + (long long)g * CH;
    float4* om = out + (long long)((n >> 3) - 1 - g) * CH;
    if (whole_wave) {
      // A lane's CH vectors are consecutive in memory, a store instruction's 64 vectors would lie 16 CH bytes apart.  The 64
      // tasks of this wavefront cover ONE contiguous 1024 CH bytes per half, so the vectors go through a 64 x CH transposition
      // first -- in the own-quarter floats these 64 tasks have just read (channel c's piece: 256 floats, nobody else's) -- and
      // leave as CH fully coalesced streaming stores per half.
      const int l = tid & 63, g0 = g - l;
      float* piece0 = smem + (nx ? (half >> 1) : 0) + 4 * g0;
      float4* rf = out + (long long)g0 * CH;                         // the forward half: groups g0 .. g0 + 63 ascending
      float4* rm = out + (long long)((n >> 3) - 1 - (g0 + 63)) * CH;  // the mirrored half: the same groups, descending
      wave_sync();  // every lane's own-quarter reads are through
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int j = l * CH + k;
        *reinterpret_cast<float4*>(piece0 + (j >> 6) * slice + 4 * (j & 63)) = make_float4(fwd[4 * k], fwd[4 * k + 1], fwd[4 * k + 2], fwd[4 * k + 3]);
      }
      wave_sync();
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(piece0 + k * slice + 4 * l);
        pcm_store4(rf + k * 64 + l, v.x, v.y, v.z, v.w);
      }
      wave_sync();
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int j = (63 - l) * CH + k;
        *reinterpret_cast<float4*>(piece0 + (j >> 6) * slice + 4 * (j & 63)) = make_float4(mir[4 * k], mir[4 * k + 1], mir[4 * k + 2], mir[4 * k + 3]);
      }
      wave_sync();
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(piece0 + k * slice + 4 * l);
        pcm_store4(rm + k * 64 + l, v.x, v.y, v.z, v.w);
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
#ifdef NVH_DIRECT8_NT
      pcm_store4(of + k, fwd[4 * k], fwd[4 * k + 1], fwd[4 * k + 2], fwd[4 * k + 3]);
      pcm_store4(om + k, mir[4 * k], mir[4 * k + 1], mir[4 * k + 2], mir[4 * k + 3]);
#else
      // (a lane's CH vectors are consecutive, a store instruction's 64 vectors are 16 CH bytes apart: plain stores, which the
      // L2 merges into whole lines -- streamed, such partial lines cost 273 us per C4 pass instead of 110)
      of[k] = make_float4(fwd[4 * k], fwd[4 * k + 1], fwd[4 * k + 2], fwd[4 * k + 3]);
      om[k] = make_float4(mir[4 * k], mir[4 * k + 1], mir[4 * k + 2], mir[4 * k + 3]);
#endif
    }
  };
  for (int t = tid; t < 2 * groups; t += NT) {
    if (groups >= 64) {  // (uniform) a wavefront's tasks belong to one overlap
      const bool nx = __builtin_amdgcn_readfirstlane((int)(t >= groups)) != 0;
      task(nx, nx ? t - groups : t, true);
    } else {
      const bool nx = t >= groups;
      task(nx, nx ? t - groups : t, false);
    }
  }
  if (A.clip) report_clipped(clipped, A.clipped_flag);
}

// The float side of one frame up to its spectra: residue adds (+ inverse coupling and the floor multiply inside the walk where the
// slab says so), else coupling and floor curves as passes of their own.  slab / spec: the frame's slab image and spectra in LDS;
// w0 .. w5, cpl_word: the slab header's words (scalar loads).  Every barrier in here is passed by the whole workgroup (the flags are
// the frame's, hence uniform).  dbgf: profiling builds' stamp row of the frame (3 walk done, 6 / 7 around the coupling passes).
template <int NT, int MAXCH, bool GENERAL>
__device__ __forceinline__ void synth_frame_spectrum(const NvhSynthArgs& A, const float* s_db, const uint32_t* s_lat, float* slab,
                                                     float* spec, unsigned w0, unsigned w1, unsigned w2, unsigned w3, unsigned w4,
                                                     unsigned w5, unsigned cpl_word, int tid, long long* dbgf = nullptr) {
#define SY_T(k) do { if (dbgf && threadIdx.x == 0) dbgf[(k)] = clock64(); } while (0)
  const int nch = A.channels;
  const int n = (int)(w0 & 0xFFFFu);
  const unsigned flags = w0 >> 24;
  const unsigned nheads = w1 & 0xFFFFu;
  const unsigned off_heads = w2 & 0xFFFFu, off_rec = w2 >> 16;
  const unsigned off_ent = w3 & 0xFFFFu;
  const unsigned lpc = w4 & 0xFFFFu, rgeom = (w4 >> 16) & 0xFFu, group = w4 >> 24;
  const unsigned lpc_magic = w5;
  const uint32_t* s_chan = reinterpret_cast<const uint32_t*>(slab) + 8;  // per channel: mode | nseg << 8 | off_seg << 16
  const int half = n >> 1;
  if ((flags & NVH_SLAB_FLOOR_FAULT) && tid == 0) atomicOr(A.err, NVH_DEVERR_FLOOR1_Y);
  // the floor curve of channel c as it lies in the slab
  auto floor_of = [&](unsigned cw, const uint4*& seg, const uint8_t*& tab) {
    const unsigned ns = (cw >> 8) & 0xFFu, oseg = cw >> 16;
    seg = reinterpret_cast<const uint4*>(slab + oseg * 4);
    tab = reinterpret_cast<const uint8_t*>(slab + (oseg + ns) * 4);
  };

  // ---- residue: one lane per GROUP of consecutive vector components of one chain, all cascade stages with the sums in
  // registers (the reference's additions in the reference's order per element) ----
  {
    const unsigned rtype = rgeom & 7u, rch = rgeom >> 4;
    const bool dig = (rgeom & NVH_SLAB_RGEOM_DIGITS) != 0;  // (uniform: the header came by scalar loads)
    const bool interleaved = !(rtype == 1 || rch == 1);  // Residue2 over several channels: component k = bin k / rch of channel k % rch
#define NVH_WALK(G, FUSE, RCH, FP)                                                                                                 \
  do {                                                                                                                             \
    if (dig) residue_walk<G, FUSE, RCH, NT, true>(slab, off_heads, off_rec, off_ent, s_lat, spec, half, nheads, lpc, lpc_magic, interleaved, flags, tid, FP); \
    else residue_walk<G, FUSE, RCH, NT, false>(slab, off_heads, off_rec, off_ent, s_lat, spec, half, nheads, lpc, lpc_magic, interleaved, flags, tid, FP, A.vq); \
  } while (0)
#ifdef NVH_ABL_NO_WALK
    if (true) {
    } else
#endif
    if (GENERAL && group == 1) {
      if (dig) residue_walk_general<NT, MAXCH, true>(slab, off_heads, off_rec, off_ent, lpc, s_lat, spec, half, tid);
      else residue_walk_general<NT, MAXCH, false>(slab, off_heads, off_rec, off_ent, lpc, s_lat, spec, half, tid, A.vq);
    } else if (MAXCH <= 2 && (flags & NVH_SLAB_FUSE_FLOOR)) {
      FloorRef F;
      const unsigned c0w = __builtin_amdgcn_readfirstlane(s_chan[0]), c1w = __builtin_amdgcn_readfirstlane(s_chan[1]);
      floor_of(c0w, F.seg[0], F.tab[0]);
      floor_of(c1w, F.seg[1], F.tab[1]);
      F.md[0] = (int)(c0w & 0xFFu); F.md[1] = (int)(c1w & 0xFFu);
      F.s_db = s_db;
      NVH_WALK(8, true, 0, &F);
    } else if (!interleaved || rch == 2) {
      if (group == 8) NVH_WALK(8, false, 0, nullptr);
      else NVH_WALK(2, false, 0, nullptr);
    } else if (MAXCH > 2 && group == 0) {
      if (dig) residue_walk_bins<NT, true>(slab, off_heads, off_rec, off_ent, lpc, lpc_magic, s_lat, spec, half, rch, tid);  // quirk B-1
      else residue_walk_bins<NT, false>(slab, off_heads, off_rec, off_ent, lpc, lpc_magic, s_lat, spec, half, rch, tid, A.vq);
    } else if (MAXCH > 2) {
      switch (rch) {  // group == 2 * rch (the slab writers)
        case 3: NVH_WALK(6, false, 3, nullptr); break;
        case 4: NVH_WALK(8, false, 4, nullptr); break;
        case 5: NVH_WALK(10, false, 5, nullptr); break;
        case 6: NVH_WALK(12, false, 6, nullptr); break;
        case 7: NVH_WALK(14, false, 7, nullptr); break;
        case 8: NVH_WALK(16, false, 8, nullptr); break;
        default: __builtin_trap();
      }
    } else {
      __builtin_trap();  // host: slab_path
    }
#undef NVH_WALK
  }
  SY_T(3);
  if (!(MAXCH <= 2 && (flags & NVH_SLAB_FUSE_FLOOR))) {
    __syncthreads();
    SY_T(6);
    if (flags & NVH_SLAB_COUPLE_PASS) {
      // inverse coupling as passes of their own (Mapping.cs:137-182), in the order the slab writers stored them (last step first)
      const unsigned cnt = cpl_word & 0xFu;
      for (unsigned k = 0; k < cnt; ++k) {
        const unsigned pr = (cpl_word >> (4 + 6 * k)) & 0x3Fu;
        float* M = spec + (pr & 7u) * (unsigned)half;
        float* An = spec + (pr >> 3) * (unsigned)half;
        for (int j = tid * 4; j < half; j += NT * 4) {
          float4 vm = *reinterpret_cast<const float4*>(M + j), va = *reinterpret_cast<const float4*>(An + j);
          couple1(vm.x, va.x); couple1(vm.y, va.y); couple1(vm.z, va.z); couple1(vm.w, va.w);
          *reinterpret_cast<float4*>(M + j) = vm;
          *reinterpret_cast<float4*>(An + j) = va;
        }
        __syncthreads();
      }
    }
    SY_T(7);
    // ---- floor multiply in place (Floor1.cs:196-222): 8 bins of one channel per lane ----
    {
      const int per_ch = half >> 3, per_sh = 31 - __clz(per_ch);  // half is a power of two
      // (three tasks of a lane side by side -- the task is a chain of dependent LDS round trips -- measured no gain in k_synth8:
      // 87.8 -> 88.4 us on C4, 54.4 -> 57.3 us on three channels)
      for (int t = tid; t < nch * per_ch; t += NT) {
        const int c = t >> per_sh, x0 = (t - (c << per_sh)) << 3;
        const unsigned cw = s_chan[c];
        const int md = (int)(cw & 0xFFu);
        if (md == 0) continue;  // the channel does not execute: its residue stays (quirk B-4)
        float* sp = spec + c * half + x0;
        float r[8];
        if (md == 1) {
          const uint4* seg; const uint8_t* tab;
          floor_of(cw, seg, tab);
          float m[8];
          *reinterpret_cast<float4*>(r) = *reinterpret_cast<const float4*>(sp);
          *reinterpret_cast<float4*>(r + 4) = *reinterpret_cast<const float4*>(sp + 4);
          floor_walk_fx<8>(seg, tab, s_db, x0, m);
#pragma unroll
          for (int q = 0; q < 8; ++q) r[q] = r[q] * m[q];
        } else if (md == 3) {
          // Floor0 (Floor0.cs:176-204): the curve's value depends on the bin's Bark section only; the host parser's thread
          // evaluated one value per section (host_slab.cpp: floor0_section_values), here the gather and the multiply
          const unsigned oseg = cw >> 16;
          const unsigned bark_off = reinterpret_cast<const uint32_t*>(slab + oseg * 4)[0];
          const float* qk = slab + (oseg + 1) * 4;
          const int32_t* __restrict__ bark = A.ipool + bark_off + x0;
          *reinterpret_cast<float4*>(r) = *reinterpret_cast<const float4*>(sp);
          *reinterpret_cast<float4*>(r + 4) = *reinterpret_cast<const float4*>(sp + 4);
          int kk[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) kk[q] = bark[q];
#pragma unroll
          for (int q = 0; q < 8; ++q) r[q] = r[q] * qk[kk[q]];
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) r[q] = 0.0f;  // Floor1.cs:218-221, Floor0.cs:208-211
        }
        *reinterpret_cast<float4*>(sp) = *reinterpret_cast<float4*>(r);
        *reinterpret_cast<float4*>(sp + 4) = *reinterpret_cast<float4*>(r + 4);
      }
    }
    // k_synth: the barrier in front of the transform sits inside imdct_wave<.., PRESYNC>; k_synth8's transform reads its
    // spectrum before the one barrier it has (WGSYNC), so the floor multiply ends with one of its own
    if (MAXCH > 2) __syncthreads();
  }
#undef SY_T
}

// ---- float side ------------------------------------------------------------------------------------------------------------
// LDS map (dynamic, floats): [ inverse_dB_table 256 | lattice pool (const_vecs * 4 - 256) | slab image cap_vecs * 4 |
//                              spectrum channels * block1 / 2 | block1 / 16 of IMDCT padding (k_synth) ]
// NT = 256, MAXCH = 2 (k_synth): mono / stereo, blocks up to 2048, 8 workgroups per CU; the floor multiply inside the chain
//   walk where the slab says so, the transform in place over the channel's own spectrum.
// NT = 512, MAXCH = 8 (k_synth8): up to eight channels (one wavefront per channel in the transform), blocks up to 4096; coupling
//   as passes of their own, the floor multiply as a pass over all channels, the transforms' slices laid over everything
//   that is dead by then (imdct_wave<.., WGSYNC>).
// MODE (k_synth only): 0 = synthesis alone, 1 = + the carried tail written by the last decoded block's workgroup, 2 = + paired
// emission.  Three instantiations, so that the launches that never emit keep the registers of the kernel that cannot (62 instead
// of 64 VGPRs at the 64-VGPR cap: 24.4 against 25.1 us for 4096 frames).
template <int NT, int MAXCH, int MODE = 0, bool GENERAL = false>
__device__ __forceinline__ void synth_body(const NvhSynthArgs& A, float* smem NVH_DBG_PARAMS) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
#ifdef NVH_ABL_EMPTY0
  if (A.f0 >= 0) return;  // (ablation build: the launch alone)
#endif
  // Which frame this workgroup takes.  Workgroups go round the eight XCDs by index (workgroup b runs on XCD b % 8, each XCD
  // with an L2 of its own), so with A.xcd_map the launch's frame list is cut into eight contiguous runs, one per XCD: frame
  // 2k of the even launch and frames 2k - 1, 2k + 1 of the odd launch then ran on the same XCD (except at the seven cuts), and
  // the quarters an emitting workgroup stages come out of that XCD's L2 instead of the memory side.
  int slot = (int)blockIdx.x;
  if (A.xcd_map) {
    const int nwg = (int)gridDim.x, q = nwg >> 3, r = nwg & 7, x = slot & 7;
    slot = x * q + (x < r ? x : r) + (slot >> 3);
  }
  const int f = A.f0 + slot * A.fstep;
  const int nch = A.channels;
  float* s_db = smem;
  const uint32_t* s_lat = reinterpret_cast<const uint32_t*>(smem + 256);
  float* slab = smem + A.const_vecs * 4;
  float* spec = slab + A.lds_vecs * 4;
  const int half_max = A.block1 >> 1;
#ifdef NVH_DEBUG
#define SY_T(k) do { if (dbg && threadIdx.x == 0) dbg[(long long)f * 24 + (k)] = clock64(); } while (0)
#else
#define SY_T(k) do { } while (0)
#endif
  SY_T(0);
#ifdef NVH_DEBUG
  if (dbg && threadIdx.x == 0) dbg[(long long)f * 24 + 22] = wall_clock64();
#endif
  // ---- one round trip: constants + the first NT * 16 bytes of the slab by LDS-DMA, the slab's header by a scalar load next to
  // them (the slabs are constant for the life of the kernel: address space 4 makes the load an s_load), the spectrum cleared
  // meanwhile; a slab beyond the speculative piece has its rest fetched as soon as the header is there -- in front of the ONE
  // barrier, not behind a second round trip ----
  const uint4* gslab = A.slabs + (long long)f * A.stride_vecs;
  typedef const __attribute__((address_space(4))) uint32_t* const_words;
  const_words gh = (const_words)(unsigned long long)gslab;
  const unsigned w0 = gh[0], w1 = gh[1], w2 = gh[2], w3 = gh[3], w4 = gh[4], w5 = gh[5], frame = gh[6], cpl_word = gh[7];
  constexpr unsigned kSpec = MAXCH > 2 ? 2 * NT : (NT < 256 ? 256 : NT);  // 16-byte units fetched before the header is known
  {
    const int v = tid;  // 16-byte unit handled by this lane: wavefront w moves units [64 w, 64 w + 64)
    constexpr int kAux = (MAXCH <= 2 && MODE < 2) ? kSlabAuxOdd : kSlabAux;
    if (v < A.cap_vecs) dma16<kAux>(gslab + v, slab + wv * 256);
    if ((MAXCH > 2 || NT < 256) && NT + v < A.cap_vecs) dma16<kAux>(gslab + NT + v, slab + (NT / 64 + wv) * 256);  // big slabs: 16 KB up front (two-wavefront workgroups: the same 4 KB)
    for (int c0 = wv * 64; c0 < A.const_vecs; c0 += NT)
      if (c0 + lane < A.const_vecs) dma16(A.consts + c0 + lane, smem + c0 * 4);
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int i = tid; i < (nch * half_max) >> 2; i += NT) reinterpret_cast<float4*>(spec)[i] = z;  // Mapping.cs:108
  }
  if constexpr (MAXCH <= 2 && MODE < 2) {
    // the even launch's slab of the frame in front of this one: one dword per 128-byte line brings it into this XCD's L2 (the
    // value is not used; the load is volatile so that it is issued)
    if (A.prefetch_prev && wv == NT / 64 - 1 && f >= 1) {
      const int lines = (A.cap_vecs < 256 ? A.cap_vecs : 256) >> 3;
      if (lane < lines) {
        const volatile unsigned* pp = reinterpret_cast<const volatile unsigned*>(A.slabs + (long long)(f - 1) * A.stride_vecs) + lane * 32;
        (void)*pp;
      }
    }
  }
  const unsigned vecs = w3 >> 16;
  if ((int)vecs > A.cap_vecs) __builtin_trap();  // host bug: the LDS slab area is sized from the batch's largest slab
  if (vecs > kSpec) {
    for (unsigned c0 = kSpec + wv * 64; c0 < vecs; c0 += NT)
      if (c0 + lane < vecs) dma16<kSlabAux>(gslab + c0 + lane, slab + c0 * 4);
  }
  __syncthreads();  // drains the DMA (vmcnt(0)) in front of the barrier
  SY_T(1);
#ifdef NVH_ABL_EMPTY1
  if (A.f0 >= 0) return;  // (ablation build: launch + the one round trip)
#endif
  const int n = (int)(w0 & 0xFFFFu);
  if (n == 0) return;
  const unsigned exec_mask = (w0 >> 16) & 0xFFu, flags = w0 >> 24;
  const uint32_t* s_chan = reinterpret_cast<const uint32_t*>(slab) + 8;  // per channel: mode | nseg << 8 | off_seg << 16
  const int half = n >> 1;
  SY_T(2);

  synth_frame_spectrum<NT, MAXCH, GENERAL>(A, s_db, s_lat, slab, spec, w0, w1, w2, w3, w4, w5, cpl_word, tid,
#ifdef NVH_DEBUG
                                           dbg ? dbg + (long long)f * 24 : nullptr
#else
                                           nullptr
#endif
  );
  SY_T(4);

  // ---- inverse MDCT (Mdct.cs:65-313), one wavefront per channel ----
  float* planes = A.work + (long long)frame * nch * A.block1;
  const int sl = (flags & NVH_SLAB_MDCT_SLOT) ? 1 : 0;
  const float* Aa = A.mdct_a[sl];
  const float* Bb = A.mdct_b[sl];
  const float* Cc = A.mdct_c[sl];
  const float* TW = A.mdct_tw[sl];
#ifdef NVH_ABL_NO_XFORM_ALL
  const bool xform = false;
#else
  const bool xform = wv < nch && ((exec_mask >> wv) & 1u);
#endif
  bool emit_self = false, emit_next = false, self_carry = false, carry_out = false;
  if constexpr (MAXCH <= 2 && MODE >= 2) {
    emit_self = A.pcm != nullptr && (flags & NVH_SLAB_EMIT_SELF);
    emit_next = A.pcm != nullptr && (flags & NVH_SLAB_EMIT_NEXT);
    self_carry = emit_self && (exec_mask & NVH_SLABX_SELF_CARRY);
  }
  if constexpr (MAXCH <= 2 && MODE >= 1) carry_out = A.carry_out != nullptr && (exec_mask & NVH_SLABX_CARRY_OUT);
  unsigned carry_window = 0u;
  if constexpr (MAXCH <= 2 && MODE >= 1) carry_window = carry_out ? __builtin_amdgcn_readfirstlane(s_chan[2]) : 0u;  // before anything overlays the slab
  // wide frames with paired emission: which overlaps this frame emits is read from its frame record (the slab header's per-channel
  // words are all taken); the direct form (synth_emit8_direct) takes the frames in the middle of the steady state
  unsigned ef8 = 0u;
  if constexpr (MAXCH > 2 && MODE >= 2) ef8 = A.pcm != nullptr ? A.frames[frame].emit_flags : 0u;  // uniform
  const bool direct8 = MAXCH > 2 && MODE >= 2 && n <= 4096 &&
                       (ef8 & (NVH_EMIT_SELF | NVH_EMIT_NEXT | NVH_EMIT_SELF_CARRY | NVH_EMIT_CARRY_OUT)) == (NVH_EMIT_SELF | NVH_EMIT_NEXT);
  if (MAXCH <= 2 && MODE >= 2 && (emit_self || emit_next)) {
    if constexpr (MAXCH <= 2 && MODE >= 2)
#ifdef NVH_DEBUG
      synth_emit<NT>(A, smem, spec, s_chan, n, nch, frame, sl, emit_self, emit_next, self_carry, carry_out, exec_mask, planes,
                     Aa, Bb, Cc, TW, tid, dbg ? dbg + (long long)f * 24 : nullptr);
#else
      synth_emit<NT>(A, smem, spec, s_chan, n, nch, frame, sl, emit_self, emit_next, self_carry, carry_out, exec_mask, planes,
                     Aa, Bb, Cc, TW, tid);
#endif
  } else
  if (MAXCH <= 2) {
    // in place over the channel's own spectrum (the transform's slice = n/2 floats + n/16 of padding: channel nch-1 spills its
    // padding past the end of the spectrum area, the one before it into the dead slab area in front of it; the workgroup
    // barrier between the floor multiply and the transform sits inside imdct_wave<.., PRESYNC>, behind the first table loads)
    if (xform) {
      const float* X = spec + wv * half;
      float* out = planes + (long long)wv * A.block1;
      float* scratch = spec + wv * half - (nch - 1 - wv) * (n >> 4);
#ifdef NVH_DEBUG
      long long* imdct_stamp = (dbg && wv == 0) ? dbg + (long long)f * 24 + 10 : nullptr;  // transform phases of channel 0's wavefront
#else
      long long* imdct_stamp = nullptr;
#endif
      switch (n) {
        case 256: imdct_wave<8, false, true, true, false, true, kSynthPF>(X, out, nullptr, scratch, Aa, Bb, Cc, TW, lane, imdct_stamp, kAblSkip); break;
        case 512: imdct_wave<9, false, true, true, false, true, kSynthPF>(X, out, nullptr, scratch, Aa, Bb, Cc, TW, lane, imdct_stamp, kAblSkip); break;
        case 1024: imdct_wave<10, false, true, true, false, true, kSynthPF>(X, out, nullptr, scratch, Aa, Bb, Cc, TW, lane, imdct_stamp, kAblSkip); break;
        case 2048: imdct_wave<11, false, true, true, false, true, kSynthPF>(X, out, nullptr, scratch, Aa, Bb, Cc, TW, lane, imdct_stamp, kAblSkip); break;
        default: __builtin_trap();  // host launches this kernel for 256 <= block0, block1 <= 2048 only
      }
    } else {
      __syncthreads();
    }
  } else {
    // Every transforming wavefront first takes its channel's whole spectrum into registers; behind the ONE workgroup barrier
    // inside imdct_wave<.., WGSYNC> the constants, the slab and all spectra are dead, and the transforms' slices (n/2 + n/16
    // floats each) are laid out back to back from the start of the LDS area.
    bool kept = false;
    if constexpr (MAXCH > 2 && MODE >= 2) kept = xform && direct8;
    if constexpr (MAXCH > 2 && MODE >= 2) if (kept) {
      const float* X = spec + wv * half;
      float* scratch = smem + wv * (half + (n >> 4));
      switch (n) {  // (blocks up to 4096: the host marks no emission beyond)
        case 256: imdct_keep_quarters<8>(X, scratch, Aa, Bb, Cc, TW, lane); break;
        case 512: imdct_keep_quarters<9>(X, scratch, Aa, Bb, Cc, TW, lane); break;
        case 1024: imdct_keep_quarters<10>(X, scratch, Aa, Bb, Cc, TW, lane); break;
        case 2048: imdct_keep_quarters<11>(X, scratch, Aa, Bb, Cc, TW, lane); break;
        case 4096: imdct_keep_quarters<12>(X, scratch, Aa, Bb, Cc, TW, lane); break;
        default: __builtin_trap();
      }
    }
    if (xform && !kept) {
      const float* X = spec + wv * half;
      float* out = planes + (long long)wv * A.block1;
      float* scratch = smem + wv * (half + (n >> 4));
      switch (n) {
        case 256: imdct_wave<8, false, true, true, true, false, true>(X, out, nullptr, scratch, Aa, Bb, Cc, TW, lane); break;
        case 512: imdct_wave<9, false, true, true, true, false, true>(X, out, nullptr, scratch, Aa, Bb, Cc, TW, lane); break;
        case 1024: imdct_wave<10, false, true, true, true, false, true>(X, out, nullptr, scratch, Aa, Bb, Cc, TW, lane); break;
        case 2048: imdct_wave<11, false, true, true, true, false, true>(X, out, nullptr, scratch, Aa, Bb, Cc, TW, lane); break;
        case 4096: imdct_wave<12, false, true, true, true, false, true>(X, out, nullptr, scratch, Aa, Bb, Cc, TW, lane); break;
        case 8192: {
          // 4096 bins per channel do not fit the registers of the in-place form: the looped transform reads the spectrum where it
          // lies and works in a slice of its own behind the spectra (mono / stereo only: the host checks the LDS budget)
          __syncthreads();  // the one barrier of the other sizes' WGSYNC form (the non-transforming wavefronts pass it below)
          float* own = spec + nch * half_max + wv * (half + (n >> 4));
          imdct_wave<13, false, true>(X, out, nullptr, own, Aa, Bb, Cc, TW, lane);
          break;
        }
        default: __builtin_trap();  // host launches this kernel for block sizes up to 8192 only
      }
    }
  }
  if (!xform && wv < nch) {
    // Mapping.cs:192-196: the residue stays in [0, n/2) (k_ola_compact windows it); its tail quarter is zero
    // (copied out before the barrier the transforming wavefronts pass: their slices may overlay this spectrum behind it)
    const float* X = spec + wv * half;
    float* out = planes + (long long)wv * A.block1;
    for (int i = lane * 4; i < half; i += 256) *reinterpret_cast<float4*>(out + i) = *reinterpret_cast<const float4*>(X + i);
    for (int i = lane * 4; i < (half >> 1); i += 256) *reinterpret_cast<float4*>(out + half + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  if (MAXCH > 2 && !xform) __syncthreads();  // the one barrier every transforming wavefront passes inside imdct_wave<.., WGSYNC>
  if constexpr (MAXCH > 2 && MODE >= 2) {
    // paired emission for wide frames: which overlaps this frame emits is read from its frame record (the slab header's
    // per-channel words are all taken); every wavefront's plane stores are complete behind the barrier
    const unsigned ef = ef8;
    if (direct8) {
      __syncthreads();  // every channel's own quarters are in its slice
      switch (nch) {
        case 1: synth_emit8_direct<NT, 1>(A, smem, n, frame, tid); break;  // (mono / stereo land here with blocks beyond 2048)
        case 2: synth_emit8_direct<NT, 2>(A, smem, n, frame, tid); break;
        case 3: synth_emit8_direct<NT, 3>(A, smem, n, frame, tid); break;
        case 4: synth_emit8_direct<NT, 4>(A, smem, n, frame, tid); break;
        case 5: synth_emit8_direct<NT, 5>(A, smem, n, frame, tid); break;
        case 6: synth_emit8_direct<NT, 6>(A, smem, n, frame, tid); break;
        case 7: synth_emit8_direct<NT, 7>(A, smem, n, frame, tid); break;
        case 8: synth_emit8_direct<NT, 8>(A, smem, n, frame, tid); break;
        default: __builtin_trap();
      }
    } else
    if (ef & (NVH_EMIT_SELF | NVH_EMIT_NEXT | NVH_EMIT_CARRY_OUT)) {
      __syncthreads();
      if ((ef & NVH_EMIT_CARRY_OUT) && A.carry_out) synth_carry_out8<NT>(A, planes, n, nch, exec_mask, A.frames[frame].window_off, tid);
      if (ef & (NVH_EMIT_SELF | NVH_EMIT_NEXT)) synth_emit8<NT>(A, smem, n, nch, frame, ef, exec_mask, tid);
    }
  }
  if constexpr (MAXCH > 2 && MODE < 2) {
    // the odd launch of a wide batch with paired emission: the last decoded block may be odd
    if (A.carry_out != nullptr && (A.frames[frame].emit_flags & NVH_EMIT_CARRY_OUT)) {  // uniform
      __syncthreads();
      synth_carry_out8<NT>(A, planes, n, nch, exec_mask, A.frames[frame].window_off, tid);
    }
  }
  if constexpr (MAXCH <= 2 && MODE >= 1) {
    if (carry_out && !(emit_self || emit_next)) {  // (an emitting frame has done it inside synth_emit)
      __syncthreads();  // every wavefront's plane stores are complete
      synth_carry_out<NT>(A, planes, n, nch, exec_mask, carry_window, tid);
    }
  }
  SY_T(5);
#ifdef NVH_DEBUG
  if (dbg && threadIdx.x == 0) dbg[(long long)f * 24 + 23] = wall_clock64();
#endif
#undef SY_T
}

// ---- frame groups (round 6): FPW consecutive frames per workgroup, the overlaps between them on chip -----------------------------
// Paired emission with one frame per workgroup sends every second frame through HBM: the odd frames' planes go out (two quarters
// per channel) and come back in through the even frames' staging -- 16.8 MB out and 16.8 MB back per 4096-frame pass of C2, a
// quarter of the pass's real traffic, on a pass that runs at 85 % of the box's copy rate under three streams.  Here a workgroup takes
// FPW consecutive frames (one wavefront per (frame, channel) in the transform: all of them busy in the longest phase, where
// k_synth has two of four idle), keeps every transform's two independent quarters in registers, puts them into the channel's dead
// slice, and overlap-adds the FPW - 1 overlaps INSIDE the group from LDS (Mode.cs:160-166 windows, StreamDecoder.cs:532-541 adds,
// :391-415 / Utils.cs:30-43 interleave + clip: synth_emit's arithmetic).  Only the overlaps between groups cross HBM, and they the
// way they did: the groups with an odd index run first ("export": first quarter of their first frame and third quarter of their
// last frame to the planes), the even groups second ("import": those quarters staged by LDS-DMA, both outer overlaps emitted).
// Plane traffic per pass: 1 / FPW of the one-frame form's.
// The emission flags keep their meaning (nvh_format.h): SELF = this frame's workgroup emits the frame's PCM (its first half over
// frame f - 1's second half), NEXT = it emits frame f + 1's; the host sets them by position in the group (nvh_launch.hip), the
// device withdraws them by the execute flags (k_parse_links) exactly as before.  A quarter goes to the plane unless the overlap it
// belongs to is emitted by this workgroup -- k_ola_compact (frames outside the steady state) and the other launch find it there.
// LDS map (floats), two phases over the same bytes (nvh_launch.hip: slab_lds_bytes sizes the larger + the table):
//   walk:      [ constants | FPW slab images (A.lds_vecs * 4 each) | FPW x channels x block1/2 spectra ]
//   transform: [ staged quarters, channels x block1/2: B(first - 1) at 0, A(last + 1) at channels * block1/4 | FPW x channels slices of
//                block1/2 + block1/16 ] -- the slices overlay dead slabs AND live spectra: every wavefront takes its spectrum into
//                registers in front of ONE LDS-only workgroup barrier (imdct_wave_sink<.., WGSYNC, .., LDSBAR>); the staging DMA,
//                issued in front of it, lands below the spectra (the host sizes the slab areas for that) and is waited for behind the
//                transforms
//   behind both: 8 (2 FPW + 1) words: the overlaps' parameters, and those of what each frame emits alone.
// Against private pads around every frame's spectra: 28.1 -> 26.7 KB for C2's shape = six workgroups per CU instead of five.
// (Tried and removed, round 6: two frames per workgroup of EIGHT wavefronts, every 256 threads walking their own frame side by side --
// k_synth's per-frame parallelism, 4 workgroups = 32 wavefronts per CU at 64 VGPRs: 171 M frames/s over three streams against 202 M for
// this form, 124 M against 123 M on one.)
template <int NT, int FPW>
__device__ __forceinline__ void synth_group_body(const NvhSynthArgs& A, float* smem NVH_DBG_PARAMS) {
  static_assert(NT / 64 >= 2 * FPW, "one wavefront per (frame, channel) in the transform");
  // (the wavefront's index through readfirstlane: what depends on it alone -- which frame and channel it transforms, where its
  // planes lie -- then lives in scalar registers)
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nch = A.channels;
  const int fa = A.f0 + (int)blockIdx.x * A.fstep;  // the group's first frame
#ifdef NVH_DEBUG
  // profiling build: shader-clock stamps of thread 0 in the row of the group's first frame (tools/dbg_phase_group.py): 0 entry,
  // 1 slabs + constants arrived, 2 / 3 the frames' walks, 4 table + barrier, 5 transform of wavefront 0, 6 staging drained, 7 emitted
#define GR_T(k) do { if (dbg && tid == 0 && fa < A.nframes) dbg[(long long)fa * 24 + (k)] = clock64(); } while (0)
#else
#define GR_T(k) do { } while (0)
#endif
  GR_T(0);
  const int half_max = A.block1 >> 1, pad_max = A.block1 >> 4;
  float* s_db = smem;
  const uint32_t* s_lat = reinterpret_cast<const uint32_t*>(smem + 256);
  float* slab0 = smem + A.const_vecs * 4;
  const int slab_words = A.lds_vecs * 4;
  float* spec0 = slab0 + FPW * slab_words;
  const int region = nch * half_max;                 // a frame's spectra
  const int slice_words = half_max + pad_max;        // a transform's slice
  float* slice0 = smem + nch * half_max;             // behind the staged quarters
  const int walk_end = A.const_vecs * 4 + FPW * (slab_words + region), xform_end = nch * half_max + FPW * nch * slice_words;
  uint32_t* otab = reinterpret_cast<uint32_t*>(smem + (walk_end > xform_end ? walk_end : xform_end));
  typedef const __attribute__((address_space(4))) uint32_t* const_words;
  // ---- one round trip: constants, the first NT * 16 bytes of every slab, the headers' size words by scalar loads; spectra
  // cleared meanwhile ----
  unsigned w0[FPW], w3[FPW];
#pragma unroll
  for (int k = 0; k < FPW; ++k) {
    w0[k] = 0u; w3[k] = 0u;
    if (fa + k < A.nframes) {  // (uniform)
      const uint4* gslab = A.slabs + (long long)(fa + k) * A.stride_vecs;
      const_words gh = (const_words)(unsigned long long)gslab;
      w0[k] = gh[0]; w3[k] = gh[3];
      if (tid < A.cap_vecs) dma16<kSlabAux>(gslab + tid, slab0 + k * slab_words + wv * 256);
    }
  }
  for (int c0 = wv * 64; c0 < A.const_vecs; c0 += NT)
    if (c0 + lane < A.const_vecs) dma16(A.consts + c0 + lane, smem + c0 * 4);
  {
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < FPW; ++k) {
      float4* sp = reinterpret_cast<float4*>(spec0 + k * region);
      for (int i = tid; i < (nch * half_max) >> 2; i += NT) sp[i] = z;  // Mapping.cs:108
    }
  }
  int nn[FPW];
  bool any = false;
#pragma unroll
  for (int k = 0; k < FPW; ++k) {
    nn[k] = (int)(w0[k] & 0xFFFFu);
    any = any || nn[k] != 0;
    const unsigned vecs = w3[k] >> 16;
    if ((int)vecs > A.cap_vecs) __builtin_trap();  // host bug: the LDS slab areas are sized from the batch's largest slab
    if (vecs > (unsigned)NT) {
      const uint4* gslab = A.slabs + (long long)(fa + k) * A.stride_vecs;
      for (unsigned c0 = NT + wv * 64; c0 < vecs; c0 += NT)
        if (c0 + lane < vecs) dma16<kSlabAux>(gslab + c0 + lane, slab0 + k * slab_words + c0 * 4);
    }
  }
  // (opt-in, NVH_GROUP_PREFETCH) the other launch's slabs of the group in front of this one, one dword per 128-byte line
  if (A.prefetch_prev && wv == NT / 64 - 1 && fa >= FPW) {
    const int lines = (A.cap_vecs < 256 ? A.cap_vecs : 256) >> 3;
#pragma unroll
    for (int k = 0; k < FPW; ++k)
      if (lane < lines) {
        const volatile unsigned* pp = reinterpret_cast<const volatile unsigned*>(A.slabs + (long long)(fa - FPW + k) * A.stride_vecs) + lane * 32;
        (void)*pp;
      }
  }
  __syncthreads();  // drains the DMA (vmcnt(0)) in front of the barrier
  if (!any) return;
  GR_T(1);

  // a slab's header word i, from its LDS image (uniform)
  auto hdr = [&](int k, int i) { return __builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t*>(slab0 + k * slab_words)[i]); };
  // ---- spectra, frame by frame (every barrier inside is the whole workgroup's) ----
  {
    // (two frames of the common shape -- digit form, stereo Residue2, fused floor multiply -- walk together: residue_walk_two)
    bool together = false;
    if constexpr (FPW == 2) {
      WalkFrame W[2];
      together = A.walk_two != 0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const unsigned w1 = hdr(k, 1), w2 = hdr(k, 2), w4 = hdr(k, 4), flags = w0[k] >> 24, rgeom = (w4 >> 16) & 0xFFu;
        W[k].slab = slab0 + k * slab_words;
        W[k].nheads = w1 & 0xFFFFu; W[k].off_heads = w2 & 0xFFFFu; W[k].off_rec = w2 >> 16; W[k].off_ent = w3[k] & 0xFFFFu;
        W[k].lpc = w4 & 0xFFFFu; W[k].lpc_magic = hdr(k, 5); W[k].flags = flags;
        W[k].spec = spec0 + k * region; W[k].half = nn[k] >> 1;
        together = together && nn[k] != 0 && (flags & NVH_SLAB_FUSE_FLOOR) && !(flags & NVH_SLAB_FLOOR_FAULT) &&
                   rgeom == (2u | NVH_SLAB_RGEOM_DIGITS | (2u << 4)) && (w4 >> 24) == 8u && W[k].nheads * W[k].lpc != 0u;
        const uint32_t* s_chan = reinterpret_cast<const uint32_t*>(W[k].slab) + 8;
        const unsigned c0w = __builtin_amdgcn_readfirstlane(s_chan[0]), c1w = __builtin_amdgcn_readfirstlane(s_chan[1]);
        const unsigned cw[2] = {c0w, c1w};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const unsigned ns = (cw[c] >> 8) & 0xFFu, oseg = cw[c] >> 16;
          W[k].F.seg[c] = reinterpret_cast<const uint4*>(W[k].slab + oseg * 4);
          W[k].F.tab[c] = reinterpret_cast<const uint8_t*>(W[k].slab + (oseg + ns) * 4);
          W[k].F.md[c] = (int)(cw[c] & 0xFFu);
        }
        W[k].F.s_db = s_db;
      }
      if (together) residue_walk_two<NT>(W, s_lat, tid);
      if (together) { GR_T(2); GR_T(3); }
    }
    if (!together)
#pragma unroll
    for (int k = 0; k < FPW; ++k) {
      if (nn[k] != 0)
        synth_frame_spectrum<NT, 2, false>(A, s_db, s_lat, slab0 + k * slab_words, spec0 + k * region, w0[k], hdr(k, 1), hdr(k, 2),
                                           w3[k], hdr(k, 4), hdr(k, 5), hdr(k, 7), tid);
      if (k < 2) GR_T(2 + k);
    }
  }

  // ---- what this workgroup emits: overlap j = the PCM of the group's frame j (j = FPW: of the frame behind the group) ----
  const bool emit = A.pcm != nullptr;
  bool self_k[FPW], next_k[FPW], cout_k[FPW], done_k[FPW];
  bool self_carry = false;
  // geometry (NVH_SLAB_GEO): block sizes of the frame before and behind, this frame's start and valid
  int gprev[FPW], gnext[FPW], gstart[FPW], gvalid[FPW];
#pragma unroll
  for (int k = 0; k < FPW; ++k) {
    const unsigned flags = w0[k] >> 24, xm = (w0[k] >> 16) & 0xFFu;
    self_k[k] = emit && nn[k] != 0 && (flags & NVH_SLAB_EMIT_SELF);
    next_k[k] = emit && nn[k] != 0 && (flags & NVH_SLAB_EMIT_NEXT);
    done_k[k] = emit && nn[k] != 0 && (xm & NVH_SLABX_DONE);
    cout_k[k] = A.carry_out != nullptr && nn[k] != 0 && (xm & NVH_SLABX_CARRY_OUT);
    if (k == 0) self_carry = self_k[0] && (xm & NVH_SLABX_SELF_CARRY);
    const unsigned geo = (self_k[k] || next_k[k] || done_k[k]) ? hdr(k, 13) : 0u;
    gprev[k] = (int)(geo & 0xFFu) << 6; gnext[k] = (int)((geo >> 8) & 0xFFu) << 6;
    gstart[k] = (int)((geo >> 16) & 0xFFu) << 6; gvalid[k] = (int)(geo >> 24) << 6;
  }
  bool act[FPW + 1];
  act[0] = self_k[0] && !self_carry;
#pragma unroll
  for (int j = 1; j < FPW; ++j) act[j] = next_k[j - 1] && self_k[j] && nn[j] == gnext[j - 1];
  act[FPW] = next_k[FPW - 1];
  // overlap j spans mov[j] = min(block sizes) samples' worth of quarters: m / 4 values of each block (ola_sym on m)
  int mov[FPW + 1];
  mov[0] = nn[0] < gprev[0] ? nn[0] : gprev[0];
#pragma unroll
  for (int j = 1; j <= FPW; ++j) mov[j] = nn[j - 1] < gnext[j - 1] ? nn[j - 1] : gnext[j - 1];
  // what a frame emits alone (the flat parts of a long window next to a short block): R1 = [start + m/2, n/2) out of its first
  // quarter (mirrored, negated), R2 = [n/2, valid) out of its third quarter
  int r1len[FPW], r2len[FPW];
#pragma unroll
  for (int k = 0; k < FPW; ++k) {
    const int mk = nn[k] < gprev[k] ? nn[k] : gprev[k];
    r1len[k] = done_k[k] ? (nn[k] >> 1) - gstart[k] - (mk >> 1) : 0;
    r2len[k] = done_k[k] ? gvalid[k] - (nn[k] >> 1) : 0;
    if (r1len[k] < 0 || r2len[k] < 0 || r1len[k] > (nn[k] >> 2) || r2len[k] > (nn[k] >> 2)) __builtin_trap();  // host bug
  }
  // The overlaps' parameters, out of the slabs before the staging overwrites them, into a table behind the spectra (one row of
  // eight words per overlap, written by lane j of the first wavefront, read by every task of the overlap: a broadcast read instead of
  // select chains over values that would stay live through the transforms): window, the earlier block's window, output position, block
  // size, float offsets from smem and channel strides of the later block's first quarter and the earlier block's third quarter.
  // Overlap 0 takes the first frame's SELF fields (chan[2], [3], [6]), overlap j >= 1 frame j - 1's NEXT fields (chan[4], [5], [7]).
  const int S0 = (int)(slice0 - smem), SK = nch * slice_words;  // float offset of frame 0's slices from smem, slice words per frame
  if (tid <= FPW) {
    // row j: window of the later block at its start, window of the earlier block at its valid, output position, m |
    // float offset from smem and channel stride of the later block's quarter values (the last m/4 of its first quarter), of the
    // earlier block's (the last m/4 of its third quarter)
    const int j = tid, kb = j == 0 ? 0 : j - 1;
    const uint32_t* H = reinterpret_cast<const uint32_t*>(slab0 + kb * slab_words);
    const int n = (int)(H[0] & 0xFFFFu);
    const unsigned geo = H[13];
    uint4 r0, r1;
    if (j == 0) {  // SELF of the first frame: A(first) own, B(first - 1) staged (m/4 values per channel)
      const int pn = (int)(geo & 0xFFu) << 6, m = n < pn ? n : pn, start = (int)((geo >> 16) & 0xFFu) << 6;
      r0.x = H[10] + (unsigned)start; r0.y = H[11] + (unsigned)(3 * (pn >> 2) - (m >> 2)); r0.z = H[14]; r0.w = (unsigned)m;
      r1.x = (unsigned)(S0 + (n >> 2) - (m >> 2)); r1.y = (unsigned)slice_words; r1.z = 0u; r1.w = (unsigned)(m >> 2);
    } else {       // NEXT of frame j - 1: B(j - 1) own; A(j) own, or A(last + 1) staged
      const int cn = (int)((geo >> 8) & 0xFFu) << 6, m = n < cn ? n : cn;
      r0.x = H[12] + (unsigned)((cn >> 2) - (m >> 2)); r0.y = H[10] + (unsigned)(3 * (n >> 2) - (m >> 2)); r0.z = H[15]; r0.w = (unsigned)m;
      r1.z = (unsigned)(S0 + kb * SK + (n >> 1) - (m >> 2)); r1.w = (unsigned)slice_words;
      if (j < FPW) { r1.x = (unsigned)(S0 + j * SK + (cn >> 2) - (m >> 2)); r1.y = (unsigned)slice_words; }
      else { r1.x = (unsigned)(nch * (A.block1 >> 2)); r1.y = (unsigned)(m >> 2); }
    }
    reinterpret_cast<uint4*>(otab)[2 * j] = r0;
    reinterpret_cast<uint4*>(otab)[2 * j + 1] = r1;
  } else if (tid <= 2 * FPW) {
    // row FPW + 1 + k: what frame k emits alone -- window, output position, block size, float offset of its slices | first index of
    // R1, samples of R1, of R2, the frame's start
    const int k = tid - FPW - 1;
    const uint32_t* H = reinterpret_cast<const uint32_t*>(slab0 + k * slab_words);
    const int n = (int)(H[0] & 0xFFFFu);
    const unsigned geo = H[13];
    const int pn = (int)(geo & 0xFFu) << 6, m = n < pn ? n : pn, start = (int)((geo >> 16) & 0xFFu) << 6, valid = (int)(geo >> 24) << 6;
    uint4 r0, r1;
    r0.x = H[10]; r0.y = H[14]; r0.z = (unsigned)n; r0.w = (unsigned)(S0 + k * SK);
    r1.x = (unsigned)(start + (m >> 1)); r1.y = (unsigned)((n >> 1) - start - (m >> 1)); r1.z = (unsigned)(valid - (n >> 1)); r1.w = (unsigned)start;
    reinterpret_cast<uint4*>(otab)[2 * tid] = r0;
    reinterpret_cast<uint4*>(otab)[2 * tid + 1] = r1;
  }
  unsigned cwin[FPW];  // a frame that writes the carried tail: its window (chan[2])
#pragma unroll
  for (int k = 0; k < FPW; ++k) cwin[k] = cout_k[k] ? hdr(k, 10) : 0u;
  __syncthreads();  // the walks are through everywhere: constants and slabs are dead, the spectra complete
  GR_T(4);

  // ---- stage the neighbours' quarters (import groups): B(first - 1), A(last + 1) ----
  float* stageB = smem;
  float* stageA = smem + nch * (A.block1 >> 2);
  {
    auto stage = [&](const float* plane0, int n, float* dst) __attribute__((always_inline)) {  // per channel n / 16 16-byte units at plane0 + c * block1
      const int q16 = n >> 4, sh = 27 - __clz(n), units = nch * q16;  // q16 = 1 << sh
      for (int u0 = wv * 64; u0 < units; u0 += NT) {
        const int u = u0 + lane;
        if (u < units) {
          const int c = u >> sh, r = u & (q16 - 1);
          dma16<kStageAux>(reinterpret_cast<const uint4*>(plane0 + (long long)c * A.block1 + 4 * r), dst + 4 * u0);
        }
      }
    };
    // (the last m/4 values of B(first - 1), of A(last + 1): all an overlap of m/2 samples reads)
    if (act[0]) stage(A.work + (long long)(fa - 1) * nch * A.block1 + 3 * (gprev[0] >> 2) - (mov[0] >> 2), mov[0], stageB);
    if (act[FPW]) stage(A.work + (long long)(fa + FPW) * nch * A.block1 + (gnext[FPW - 1] >> 2) - (mov[FPW] >> 2), mov[FPW], stageA);
  }

  // ---- inverse MDCT (Mdct.cs:65-313): one wavefront per (frame, channel), the two independent quarters stay in registers; a
  // quarter goes to the plane unless its overlap is emitted here ----
  {
    const int k = wv >> 1, c = wv & 1;
    const bool tw = k < FPW;
    int n = 0;
    unsigned mw0 = 0u;
    bool wrA = true, wrB = true, keep = false;
#pragma unroll
    for (int kk = 0; kk < FPW; ++kk)
      if (kk == k) {
        n = nn[kk]; mw0 = w0[kk];
        const bool a_here = kk == 0 ? self_k[0] : act[kk], b_here = kk == FPW - 1 ? next_k[FPW - 1] : act[kk + 1];
        // (a frame k_ola_compact emits is read there quarter by quarter, whatever its neighbours do)
        wrA = !a_here || cout_k[kk] || !done_k[kk];
        wrB = !b_here || cout_k[kk] || !done_k[kk];
        keep = a_here || b_here || done_k[kk];
      }
    const unsigned exec_mask = (mw0 >> 16) & 0xFFu, flags = mw0 >> 24;
    const int half = n >> 1;
    const int sl = (flags & NVH_SLAB_MDCT_SLOT) ? 1 : 0;
    const float* X = spec0 + k * region + c * half;
    float* slice = slice0 + (k * nch + c) * slice_words;
    float* plane = A.work + ((long long)(fa + k) * nch + c) * A.block1;  // (a batch with paired emission has its slabs in frame order)
    const bool mine = tw && c < nch && n != 0;
    if (mine && !((exec_mask >> c) & 1u)) {
      // Mapping.cs:192-196: the residue stays in [0, n/2) (k_ola_compact windows it); its tail quarter is zero
      // (copied out in front of the barrier: behind it the slices overlay this spectrum)
      for (int i = lane * 4; i < half; i += 256) *reinterpret_cast<float4*>(plane + i) = *reinterpret_cast<const float4*>(X + i);
      for (int i = lane * 4; i < (half >> 1); i += 256) *reinterpret_cast<float4*>(plane + half + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (mine && ((exec_mask >> c) & 1u)) {
      float4 ca[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, cb[2] = {ca[0], ca[0]};
      auto sink = [&](int slot, int, float4 v) {  // slot = 4 h + q: q = 0 first quarter, q = 2 third quarter (1, 3: their mirrors)
        if (slot == 0) ca[0] = v; else if (slot == 2) cb[0] = v; else if (slot == 4) ca[1] = v; else if (slot == 6) cb[1] = v;
      };
      const float* Aa = A.mdct_a[sl];
      const float* Bb = A.mdct_b[sl];
      const float* Cc = A.mdct_c[sl];
      const float* TW = A.mdct_tw[sl];
      switch (n) {
        case 256: imdct_wave_sink<8, false, decltype(sink), true, true, false, kSynthPFEmit, true>(X, nullptr, slice, Aa, Bb, Cc, TW, lane, sink, nullptr, kAblSkip); break;
        case 512: imdct_wave_sink<9, false, decltype(sink), true, true, false, kSynthPFEmit, true>(X, nullptr, slice, Aa, Bb, Cc, TW, lane, sink, nullptr, kAblSkip); break;
        case 1024: imdct_wave_sink<10, false, decltype(sink), true, true, false, kSynthPFEmit, true>(X, nullptr, slice, Aa, Bb, Cc, TW, lane, sink, nullptr, kAblSkip); break;
        case 2048: imdct_wave_sink<11, false, decltype(sink), true, true, false, kSynthPFEmit, true>(X, nullptr, slice, Aa, Bb, Cc, TW, lane, sink, nullptr, kAblSkip); break;
        default: __builtin_trap();  // host launches this kernel for 256 <= block0, block1 <= 2048 only
      }
      if (lane < (n >> 5)) {  // lanes with output: four values of each quarter at 4 i8, i8 = lane and i8 = n/16 - 1 - lane
        const int i8a = lane, i8b = (n >> 4) - 1 - lane;
        if (wrA) {
          *reinterpret_cast<float4*>(plane + 4 * i8a) = ca[0];
          *reinterpret_cast<float4*>(plane + 4 * i8b) = ca[1];
        }
        if (wrB) {
          *reinterpret_cast<float4*>(plane + half + 4 * i8a) = cb[0];
          *reinterpret_cast<float4*>(plane + half + 4 * i8b) = cb[1];
        }
        if (keep) {  // the channel's own quarters, for the overlap-adds below: A in [0, n/4), B in [n/4, n/2) of its dead slice
          float* own = slice;
          *reinterpret_cast<float4*>(own + 4 * i8a) = ca[0];
          *reinterpret_cast<float4*>(own + 4 * i8b) = ca[1];
          *reinterpret_cast<float4*>(own + (half >> 1) + 4 * i8a) = cb[0];
          *reinterpret_cast<float4*>(own + (half >> 1) + 4 * i8b) = cb[1];
        }
      }
    } else {
      lds_barrier();  // the one barrier every transforming wavefront passes inside imdct_wave_sink<.., WGSYNC, .., LDSBAR>
    }
  }
  GR_T(5);
  __syncthreads();  // drains the staging DMA; every own quarter is in LDS, every plane store of the workgroup is issued
  GR_T(6);

#pragma unroll
  for (int k = 0; k < FPW; ++k)
    if (cout_k[k])  // the block that becomes the next batch's carried tail (its whole plane was written above)
      synth_carry_out<NT>(A, A.work + (long long)(fa + k) * nch * A.block1, nn[k], nch, (w0[k] >> 16) & 0xFFu, cwin[k], tid);
  if (self_carry)  // the batch's first frame
    synth_self_carry<NT>(A, slice0, nn[0], nch, __builtin_amdgcn_readfirstlane(otab[0]), __builtin_amdgcn_readfirstlane(otab[2]), tid, slice_words);

  // ---- overlap-add + interleave + clip: lane task = (overlap j, group of four compact indices i0); it produces sample times
  // i0 .. i0 + 3 and n/2 - 4 - i0 .. n/2 - 1 - i0 of every channel (kernels.hip: ola_sym) ----
  int cnt[FPW + 2];
  cnt[0] = 0;
#pragma unroll
  for (int j = 0; j <= FPW; ++j) cnt[j + 1] = cnt[j] + (act[j] ? (mov[j] >> 4) : 0);
  int clipped = 0;
  for (int t = tid; t < cnt[FPW + 1]; t += NT) {
    int j = 0;
#pragma unroll
    for (int jj = 1; jj <= FPW; ++jj) j += (int)(t >= cnt[jj]);
    int g = t;
#pragma unroll
    for (int jj = 1; jj <= FPW; ++jj) g = j == jj ? t - cnt[jj] : g;
    const uint4 r0 = reinterpret_cast<const uint4*>(otab)[2 * j], r1 = reinterpret_cast<const uint4*>(otab)[2 * j + 1];
    const int n = (int)r0.w, half = n >> 1, i0 = 4 * g;  // (n: the overlap's m)
    const float* __restrict__ w = A.windows + r0.x;
    const float* __restrict__ wp = A.windows + r0.y;
    const float4 wf = *reinterpret_cast<const float4*>(w + i0);
    const float4 wm = *reinterpret_cast<const float4*>(w + (half - 4 - i0));
    const float4 pf = *reinterpret_cast<const float4*>(wp + i0);
    const float4 pm = *reinterpret_cast<const float4*>(wp + (half - 4 - i0));
    float fwd[8], mir[8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < nch) {
        const float4 a = *reinterpret_cast<const float4*>(smem + r1.x + c * r1.y + i0);
        const float4 b = *reinterpret_cast<const float4*>(smem + r1.z + c * r1.w + i0);
        float4 v = make_float4(a.x * wf.x, a.y * wf.y, a.z * wf.z, a.w * wf.w);
        const float4 tt = make_float4(b.x * pf.x, b.y * pf.y, b.z * pf.z, b.w * pf.w);
        v.x = v.x + tt.x; v.y = v.y + tt.y; v.z = v.z + tt.z; v.w = v.w + tt.w;
        float4 u = make_float4(-a.w * wm.x, -a.z * wm.y, -a.y * wm.z, -a.x * wm.w);
        const float4 r = make_float4(b.w * pm.x, b.z * pm.y, b.y * pm.z, b.x * pm.w);
        u.x = u.x + r.x; u.y = u.y + r.y; u.z = u.z + r.z; u.w = u.w + r.w;
        if (A.clip) {
          v.x = clip_value(v.x, &clipped); v.y = clip_value(v.y, &clipped);
          v.z = clip_value(v.z, &clipped); v.w = clip_value(v.w, &clipped);
          u.x = clip_value(u.x, &clipped); u.y = clip_value(u.y, &clipped);
          u.z = clip_value(u.z, &clipped); u.w = clip_value(u.w, &clipped);
        }
        fwd[c] = v.x; fwd[2 + c] = v.y; fwd[4 + c] = v.z; fwd[6 + c] = v.w;
        mir[c] = u.x; mir[2 + c] = u.y; mir[4 + c] = u.z; mir[6 + c] = u.w;
      }
    }
    float* out = A.pcm + (long long)r0.z * nch;
    if (nch == 2) {
      float4* of = reinterpret_cast<float4*>(out) + 2 * (long long)g;
      float4* om = reinterpret_cast<float4*>(out) + 2 * (long long)((n >> 3) - 1 - g);
      pcm_store4(of, fwd[0], fwd[1], fwd[2], fwd[3]);
      pcm_store4(of + 1, fwd[4], fwd[5], fwd[6], fwd[7]);
      pcm_store4(om, mir[0], mir[1], mir[2], mir[3]);
      pcm_store4(om + 1, mir[4], mir[5], mir[6], mir[7]);
    } else {
      pcm_store4(reinterpret_cast<float4*>(out) + g, fwd[0], fwd[2], fwd[4], fwd[6]);
      pcm_store4(reinterpret_cast<float4*>(out) + ((n >> 3) - 1 - g), mir[0], mir[2], mir[4], mir[6]);
    }
  }
  // ---- what a frame emits alone: lane task = four consecutive sample times of R1 or R2 of one frame (window multiply of
  // Mode.cs:160-166 on the block's own values, nothing overlapped onto them; kernels.hip: compact_value) ----
  int fcnt[FPW + 1];
  fcnt[0] = 0;
#pragma unroll
  for (int k = 0; k < FPW; ++k) fcnt[k + 1] = fcnt[k] + ((r1len[k] + r2len[k]) >> 2);
  for (int t = tid; t < fcnt[FPW]; t += NT) {
    int k = 0;
#pragma unroll
    for (int kk = 1; kk < FPW; ++kk) k += (int)(t >= fcnt[kk]);
    int g = t;
#pragma unroll
    for (int kk = 1; kk < FPW; ++kk) g = k == kk ? t - fcnt[kk] : g;
    const uint4 r0 = reinterpret_cast<const uint4*>(otab)[2 * (FPW + 1 + k)], r1 = reinterpret_cast<const uint4*>(otab)[2 * (FPW + 1 + k) + 1];
    const int n = (int)r0.z, q1 = (int)(r1.y >> 2);
    const bool second = g >= q1;
    const int idx0 = second ? (n >> 1) + 4 * (g - q1) : (int)r1.x + 4 * g;
    const float4 wv4 = *reinterpret_cast<const float4*>(A.windows + r0.x + idx0);
    // first half: y[idx] = -A[n/2 - 1 - idx] (idx >= n/4); second half: y[idx] = B[idx - n/2] (idx < 3n/4); B lies behind A in the slice
    const int src = second ? (n >> 2) + (idx0 - (n >> 1)) : (n >> 1) - 4 - idx0;
    float o[8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < nch) {
        const float4 x = *reinterpret_cast<const float4*>(smem + r0.w + c * slice_words + src);
        float4 v = second ? make_float4(x.x * wv4.x, x.y * wv4.y, x.z * wv4.z, x.w * wv4.w)
                          : make_float4(-x.w * wv4.x, -x.z * wv4.y, -x.y * wv4.z, -x.x * wv4.w);
        if (A.clip) {
          v.x = clip_value(v.x, &clipped); v.y = clip_value(v.y, &clipped);
          v.z = clip_value(v.z, &clipped); v.w = clip_value(v.w, &clipped);
        }
        o[c] = v.x; o[2 + c] = v.y; o[4 + c] = v.z; o[6 + c] = v.w;
      }
    }
    float* out = A.pcm + ((long long)r0.y + (idx0 - (int)r1.w)) * nch;
    if (nch == 2) {
      pcm_store4(reinterpret_cast<float4*>(out), o[0], o[1], o[2], o[3]);
      pcm_store4(reinterpret_cast<float4*>(out) + 1, o[4], o[5], o[6], o[7]);
    } else {
      pcm_store4(reinterpret_cast<float4*>(out), o[0], o[2], o[4], o[6]);
    }
  }
  if (A.clip && emit) report_clipped(clipped, A.clipped_flag);
  GR_T(7);
#undef GR_T
}

// 8 waves per SIMD = 8 resident workgroups per CU: the register budget (64 VGPRs) is part of the design
#ifndef NVH_SYNTH_NT
#define NVH_SYNTH_NT SP_THREADS
#define NVH_SYNTH_WPE 8, 8
#endif
extern "C" __global__ void __launch_bounds__(NVH_SYNTH_NT) __attribute__((amdgpu_waves_per_eu(NVH_SYNTH_WPE)))
k_synth(NvhSynthArgs A NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  synth_body<NVH_SYNTH_NT, 2>(A, smem NVH_DBG_ARGS);
}

// up to eight channels, blocks up to 4096: 8 wavefronts per workgroup, the CU's LDS decides how many are resident
extern "C" __global__ void __launch_bounds__(NVH_SYNTH_NT) __attribute__((amdgpu_waves_per_eu(NVH_SYNTH_WPE)))
k_synth_tail(NvhSynthArgs A NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  synth_body<NVH_SYNTH_NT, 2, 1>(A, smem NVH_DBG_ARGS);
}

// up to eight channels, blocks up to 4096: 8 wavefronts per workgroup, the CU's LDS decides how many are resident
extern "C" __global__ void __launch_bounds__(NVH_SYNTH_NT) __attribute__((amdgpu_waves_per_eu(NVH_SYNTH_WPE)))
k_synth_emit(NvhSynthArgs A NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  synth_body<NVH_SYNTH_NT, 2, 2>(A, smem NVH_DBG_ARGS);
}

// mono / stereo streams some of whose frames need the general bin walk (never with paired emission)
extern "C" __global__ void __launch_bounds__(SP_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_synth_g(NvhSynthArgs A NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  synth_body<SP_THREADS, 2, 0, true>(A, smem NVH_DBG_ARGS);
}

// up to eight channels, blocks up to 4096: 8 wavefronts per workgroup, the CU's LDS decides how many are resident
extern "C" __global__ void __launch_bounds__(512)
k_synth8(NvhSynthArgs A NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  synth_body<512, NVH_SLAB_MAX_CH>(A, smem NVH_DBG_ARGS);
}

// ... + the general bin walk
extern "C" __global__ void __launch_bounds__(512)
k_synth8_g(NvhSynthArgs A NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  synth_body<512, NVH_SLAB_MAX_CH, 0, true>(A, smem NVH_DBG_ARGS);
}

// the even frames of a wide batch with paired emission: + the overlap-add of the steady-state overlaps (synth_emit8)
extern "C" __global__ void __launch_bounds__(512)
k_synth8_emit(NvhSynthArgs A NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  synth_body<512, NVH_SLAB_MAX_CH, 2>(A, smem NVH_DBG_ARGS);
}

// frame groups: two frames per workgroup (four wavefronts: one per (frame, channel)); LDS, not registers, decides the residency
extern "C" __global__ void __launch_bounds__(256)
k_synth_group2(NvhSynthArgs A NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  synth_group_body<256, 2>(A, smem NVH_DBG_ARGS);
}

// ... four frames per workgroup (eight wavefronts)
extern "C" __global__ void __launch_bounds__(512)
k_synth_group4(NvhSynthArgs A NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  synth_group_body<512, 4>(A, smem NVH_DBG_ARGS);
}
