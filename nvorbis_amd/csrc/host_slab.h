// host_slab.h -- the host packet parser's output in the form the slab synthesis kernels read (nvh_format.h: NvhSlabHdr).
//
// Everything about a frame that does not depend on a float is settled at parse time, on the host thread that walked the
// packet anyway: Floor1.UnwrapPosts and the walk over the sorted, flagged posts (Floor1.cs:196-297) become the segment list
// + the per-four-bins segment table of each channel, the (stage, partition, channel) geometry of every vector write
// (Residue0.cs:132-175, Residue2.cs:23-47) becomes chain-major pair records.  k_synth / k_synth8 then fetch one contiguous
// slab per frame by LDS-DMA and start on floats with their first instruction; no integer kernel runs between the parser
// and the synthesis (round 3 had k_prepare_slabs there, once per upload).
#pragma once
#include <cstdint>
#include <vector>

#include "host_parse.h"
#include "host_setup.h"
#include "nvh_format.h"

namespace nvh {

struct SlabVec { uint32_t x, y, z, w; };  // one 16-byte unit
static_assert(sizeof(SlabVec) == 16, "slabs are addressed in 16-byte units");

// What the slab records need to know about the setup beyond nvh::Setup: the codebook directory as the device holds it
// (lattice pool offsets, reciprocal magics; nvh_setup.hip builds it).
struct SlabSetup {
  std::vector<NvhDevBook> books;
  std::vector<uint8_t> residue_b1;  // per residue: quirk B-1 aliasing only (residue_alias_b1): its frames are walked bin by bin
  // per residue: neither on the pair path nor a B-1 residue, but inside the general bin walk's contract (residue_general_ok):
  // Residue0, books of odd dimension, Residue2 over one or two channels with aliasing partitions.  Frames of such a residue,
  // and frames with more than one residue pass (several submaps), carry a group list (nvh_format.h: NvhSlabHdr::group == 1).
  std::vector<uint8_t> residue_general;
  std::vector<uint8_t> residue_pair;  // per residue: on the pair path (NvhDevResidue::pair_path)
  // the lattice pool as the device holds it (build_book_directory), kept for nvh_stream_lattice_pool: what a record's lattice
  // offset points into
  std::vector<uint32_t> lattice;
  // the VQ table pool (build_book_directory), kept for nvh_stream_vq_pool: what the one lattice-pool word of a book with an explicit
  // table points into
  std::vector<float> vq;
  // Digit form of a slab's vector entries (round 5; nvh_format.h: NVH_SLAB_RGEOM_DIGITS).  A lattice book's entry number is
  // its components' digits in base lat_values (Codebook.cs:242-260); peeling them is integer work, so the parser does it: the
  // slab carries one byte per vector component, digit * 4 = the byte offset of the component's value from the book's first word
  // in the VALUE POOL (per book: its lat_values distinct floats, then +0.0f -- the slot a "no vector was added here" component
  // points at, quirks B-14 / B-16), and the synthesis kernels' walk is a byte read, an LDS read and an add per component.
  // val_off[b]: the book's first word, counted from the start of the lattice pool (the value pool follows it in the kernels'
  // constants block); dig_tab + dig_off[b]: entries * dim bytes, the digit bytes of every entry; 0xFFFFFFFF: not a lattice
  // book of at most 63 values.  digits_ok: every book a slab residue uses has them (classify_residues).
  std::vector<uint32_t> val_pool, val_off, dig_off;
  std::vector<uint8_t> dig_tab;
  bool digits_ok = false;
  size_t pool_words = 0;  // lattice pool + value pool, in words: what a record's 12-bit offset must reach
  // Floor0: where the Bark map of floor i for block0 / block1 lies in the device's int pool (nvh_setup.hip lays the maps out in
  // floor order, block0 then block1), 0xFFFFFFFF for a Floor1
  std::vector<uint32_t> floor0_bark_off[2];
};

// A Floor0 curve takes one value per Bark section (Floor0.cs:176-204: the value depends on barkMap[i] only): q[k] for every k a
// bin of this block size maps to, evaluated here with the reference's expression shapes -- 2 cos(coeff), the p / q products in
// float, amp / (float)sqrt((double)(p + q)) - ampOfs, (float)exp((double)(q * 0.11512925f)) -- so that the kernel only gathers
// (no double precision, no device math library: the same libm as the CPU oracle's, bit for bit by construction).
// qk receives bark_map_size floats (unused k: 0); returns false when a bin maps outside wMap (the reference would throw).
bool floor0_section_values(const Floor0& f, int slot, int half, float amp, const float* coeff, float* qk);

// Quirk B-1 on its own: a Residue2 over 3..8 channels whose partitions do not start on a bin boundary (`offset /= channels`
// truncates and chPtr restarts at 0, Residue2.cs:25-27, so neighbouring partitions share a bin), every book a lattice book of
// even dimension that divides the partition, at least two bins per partition (a bin then belongs to at most two partitions).
bool residue_alias_b1(const Setup& S, const SlabSetup& X, const Residue& r);
// Components the longest vector write of a partition of residue r covers: partition_size, or -- Residue1 / Residue2 with a book whose
// dimension does not divide it (Residue1.cs:12-22: whole entries are added, the last one runs over into the next partition's
// elements) -- ceil(partition_size / dim) * dim for the worst book.  The general bin walk's `cover` comes from it.
uint32_t residue_max_span(const Setup& S, const Residue& r);

// The general bin walk (kernels_synth.hip: residue_walk_general) takes a residue when every book it uses is a lattice book whose
// dimension divides the partition size (no vector overrun; Residue0: partition_size / dimensions whole steps), partitions of 2
// ... 4096 components, and -- Residue2 over several channels with partitions off the bin grid (quirk B-1) -- at least two bins
// per partition.  The 16-bit reciprocals the walk divides with are checked over every index it can see.
bool residue_general_ok(const Setup& S, const SlabSetup& X, const Residue& r);
// The pair path (kernels_synth.hip: residue_walk): Residue1 / Residue2 without aliasing partitions, an even partition size,
// every book a lattice book of even dimension that divides it.
bool residue_pair_ok(const Setup& S, const SlabSetup& X, const Residue& r);
// Fills X.residue_pair / residue_b1 / residue_general (X.books must be there).  no_pair: the NVH_NO_PAIR A/B switch.
void classify_residues(const Setup& S, SlabSetup& X, bool no_pair);

struct SlabBatch {
  std::vector<SlabVec> data;      // the slabs back to back (frame order), each a whole number of 16-byte units
  std::vector<uint32_t> first;    // first unit of frame f's slab in `data`; first[nframes] = data.size()
  uint32_t max_vecs = 0;          // the largest slab
  void clear() { data.clear(); first.clear(); max_vecs = 0; }
};

// Fills X.books (+ the lattice pool and the VQ table pool the device image is made of; nvh_setup.hip uploads them).
void build_book_directory(const Setup& S, SlabSetup& X, std::vector<float>& vq, std::vector<uint32_t>& lattice);

// Writes the slabs of every frame of P (frames[].emit_flags decided by the caller).  The stream shape must be inside the slab
// kernels' contract (nvh_launch.hip: slab_path); returns NVH_ERR_UNSUPPORTED when a frame is not (the caller falls back to
// the descriptor kernels), NVH_OK otherwise.
int build_slabs(const Setup& S, const SlabSetup& X, const FrameBatch& P, SlabBatch& out);

// One channel's Floor1 curve as segments: UnwrapPosts + the sorted walk.  `posts` = the raw values Floor1.Unpack read
// (post_count of them).  seg receives (x | xend << 16, y, step lo, step hi) per segment; returns the segment count, sets
// *fault when a drawn value falls outside inverse_dB_table (quirk B-7).  Exposed for the unit tests.
int floor1_segments(const Floor1& f, const uint16_t* posts, int post_count, int half, SlabVec* seg, bool* fault);

}  // namespace nvh
