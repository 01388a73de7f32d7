// kernels_common.h -- device-side parameter blocks for the gfx950 synthesis kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nvh_format.h"

// Pointers into the per-stream device arena that holds the setup (uploaded once).
struct NvhDevSetup {
  int32_t channels, block0, block1, nbooks;
  const float* vq;                // VQ lookup tables of every codebook
  const uint32_t* lattice;        // lattice pool (NvhDevBook::lat_off), lattice_words entries
  int32_t lattice_words;
  int32_t fused_tail_ok;          // <= 2 channels, Floor1 only, every mapping has <= 1 coupling step (k_spectrum's fused tail)
  const NvhDevBook* books;
  const NvhDevFloor* floors;
  const NvhDevResidue* residues;
  const NvhDevMapping* mappings;
  const uint8_t* coupling;        // (magnitude, angle) byte pairs
  const float* windows;           // window pool (Mode.cs:69-100)
  const int32_t* ipool;           // Floor0 Bark maps
  const float* fpool;             // Floor0 wdel maps
  const float* mdct_a[2];         // Mdct.cs:40-55 tables for block0 / block1
  const float* mdct_b[2];
  const float* mdct_c[2];
  const uint16_t* mdct_br[2];
  const float* mdct_tw[2];        // lane-ordered copies of _a for the wavefront IMDCT (host_setup.cpp)
  const uint32_t* recip;          // recip[d] = floor((2^32 - 1) / d) for 1 <= d <= block1 / 2 (segment lengths of a floor curve)
};

// One uploaded frame batch.
struct NvhDevBatch {
  const NvhFrame* frames;
  const NvhChan* chans;
  const NvhResPass* passes;
  const NvhResOp* ops;
  const uint16_t* op_link;        // per op: next op of the same partition/channel (host_parse.h FrameBatch::op_link)
  const uint16_t* entries;
  const uint16_t* posts;
  const float* coeffs;
  int32_t nframes, pad;
};

// LDS words of one per-channel floor scratch block of k_spectrum (kernels_spectrum.hip: FloorScratch)
#define NVH_SP_FLOOR_SCRATCH_WORDS 332

// Profiling aids of the spectrum kernels (per-workgroup phase timestamps, phases masked out one at a time) exist only in
// the debug build of the library (python -m nvorbis_amd.build --debug -> libnvorbis_hip_dbg.so, -DNVH_DEBUG); the release
// kernels take neither parameter, so nothing in the shipped .so can skip work.
#ifdef NVH_DEBUG
#define NVH_DBG_PARAMS , long long* dbg, int phase_mask
#define NVH_DBG_ARGS , dbg, phase_mask
#else
#define NVH_DBG_PARAMS
#define NVH_DBG_ARGS
#endif

// error word written by kernels when the reference would have thrown (index out of range)
enum { NVH_DEVERR_FLOOR1_Y = 1, NVH_DEVERR_FLOOR0_W = 2 };

// Arguments of the slab synthesis kernel (kernels_synth.hip).
// groups of four sample times per workgroup of k_ola_compact's LDS-interleaving path (more than two channels)
#define NVH_OLA_GW 64

struct NvhSynthArgs {
  const uint4* consts;      // inverse_dB_table (256 floats) followed by the lattice pool, const_vecs 16-byte units
  const uint4* slabs;       // nframes slabs at stride_vecs
  float* work;              // [frame][channel][block1] planes: receives the compact IMDCT output k_ola_compact reads
  int* err;                 // device error word
  const float* mdct_a[2];
  const float* mdct_b[2];
  const float* mdct_c[2];
  const float* mdct_tw[2];
  const int32_t* ipool;     // Floor0 Bark maps (NvhDevSetup::ipool)
  const float* vq;          // the VQ pool (NvhDevSetup::vq): books with an explicit table are gathered from it (kernels_synth.hip: table_value)
  const NvhFrame* frames;   // the batch's frame records (k_synth8_emit reads the overlaps' windows and output positions from them)
  int const_vecs, stride_vecs, cap_vecs;  // cap_vecs: largest slab of the batch
  int lds_vecs;             // the LDS slab area (>= cap_vecs; paired emission stages the neighbours' quarters over constants + slab)
  int channels, block1;
  int f0, fstep;            // workgroup b synthesises frame f0 + b * fstep (paired emission: odd frames, then even frames)
  int nframes;              // frames of the batch (frame groups: a group's frames beyond the batch's end are skipped)
  int xcd_map;              // frames in eight contiguous runs, one per XCD (kernels_synth.hip: synth_body)
  int walk_two;             // frame groups of two: the two frames' residue walks in one loop (kernels_synth.hip: residue_walk_two)
  int prefetch_prev;        // paired emission, odd launch: workgroup b touches the slab of frame f - 1, which workgroup b of the even
                            // launch fetches next -- on the same XCD (workgroups go round the XCDs by index), so from that XCD's L2
  // paired emission (nvh_format.h: NVH_EMIT_*); pcm == nullptr: off, every frame leaves its plane for k_ola_compact
  float* pcm;
  const float* windows;
  int clip;
  int* clipped_flag;
  const float* carry;       // the carried tail this batch's first frame overlaps with (fully windowed), NVH_EMIT_SELF_CARRY
  float* carry_out;         // receives the last decoded block, fully windowed (NVH_EMIT_CARRY_OUT); nullptr: k_ola_compact writes it
};

#ifdef __HIPCC__
// PCM leaves the chip (a copy engine or the gather reads it next) and no kernel reads it again: streaming stores (`nt`), which do
// not displace what the kernels do re-read -- the odd frames' planes, the slabs the odd launch touched for the even one -- from
// the L2 / Infinity Cache.  Same box, three streams, working set past the Infinity Cache: 24.0 -> 22.2 us per 4096-frame pass.
__device__ __forceinline__ void pcm_store4(float4* p, float a, float b, float c, float d) {
  typedef float nvh_v4f __attribute__((ext_vector_type(4)));
  const nvh_v4f v = {a, b, c, d};
#ifdef NVH_PCM_POLICY
#define NVH_STR2(x) #x
#define NVH_STR(x) NVH_STR2(x)
  asm volatile("global_store_dwordx4 %0, %1, off " NVH_STR(NVH_PCM_POLICY) : : "v"(p), "v"(v) : "memory");
#else
  __builtin_nontemporal_store(v, reinterpret_cast<nvh_v4f*>(p));
#endif
}
__device__ __forceinline__ void pcm_store1(float* p, float v) { __builtin_nontemporal_store(v, p); }
// ... and the streaming load of 16 bytes that exactly one lane reads exactly once (a neighbour frame's quarter)
__device__ __forceinline__ float4 stream_load4(const float* p) {
  typedef float nvh_v4f __attribute__((ext_vector_type(4)));
  const nvh_v4f v = __builtin_nontemporal_load(reinterpret_cast<const nvh_v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// Utils.cs:30-43, without branches (two compares, two selects; the flag is an OR of the compare masks): the early-return form
// compiles to two exec-mask regions per sample.  A NaN compares false twice and passes through, as in the reference.
__device__ __forceinline__ float clip_value(float v, int* clipped) {
  const bool hi = v > .99999994f, lo = v < -.99999994f;
  *clipped |= (int)(hi | lo);
  return hi ? 0.99999994f : (lo ? -0.99999994f : v);
}

// Four consecutive positions idx0 .. idx0+3 of a block (idx0 a multiple of 4: a group never straddles a quarter) from its compact
// plane -- the two independent quarters y[0, n/4) and y[n/2, 3n/4) of the inverse MDCT, the others follow from
// y[n/2-1-x] = -y[x] and y[n-1-x] = y[n/2+x] (Mdct.cs:275-303); a channel that does not execute keeps its residue in
// [0, n/2) and zeros behind it (Mapping.cs:192-196) -- times the window (Mode.cs:160-166).
__device__ __forceinline__ float4 compact_value4(const float* __restrict__ plane, const float* __restrict__ w, int n,
                                                 int exec, int idx0) {
  const int n2 = n >> 1, n4 = n >> 2;
  float4 y;
  if (exec) {
    if (idx0 < n4 || (idx0 >= n2 && idx0 < n2 + n4)) {
      y = *reinterpret_cast<const float4*>(plane + idx0);
    } else if (idx0 < n2) {
      const float4 r = *reinterpret_cast<const float4*>(plane + (n2 - 4 - idx0));
      y = make_float4(-r.w, -r.z, -r.y, -r.x);
    } else {
      const float4 r = *reinterpret_cast<const float4*>(plane + (n + n2 - 4 - idx0));
      y = make_float4(r.w, r.z, r.y, r.x);
    }
  } else {
    y = idx0 < n2 ? *reinterpret_cast<const float4*>(plane + idx0) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float4 ww = *reinterpret_cast<const float4*>(w + idx0);
  return make_float4(y.x * ww.x, y.y * ww.y, y.z * ww.z, y.w * ww.w);
}

// HasClipped (StreamDecoder.cs:728) is sticky: one lane per wavefront that clipped looks at the flag and only sets it
// while it is still clear.  A stream that clips everywhere (loud material; Floor0 curves on random bits) otherwise
// serialises one atomic per lane -- or still 8192 per launch with one per wavefront, ~35 us -- on a single address.
__device__ __forceinline__ void report_clipped(int clipped, int* __restrict__ clipped_flag) {
  const unsigned long long any = __ballot(clipped != 0);
  if (any && (int)(threadIdx.x & 63u) == __ffsll((long long)any) - 1) {
    if (__atomic_load_n(clipped_flag, __ATOMIC_RELAXED) == 0) atomicOr(clipped_flag, 1);
  }
}
#endif
