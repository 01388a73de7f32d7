// nvh_format.h -- plain-old-data layouts shared by the host bit-parser and the gfx950 kernels.
//
// Two kinds of data cross from host to HBM:
//   * the SETUP (once per stream): codebook VQ tables, floor / residue / mapping / mode parameters,
//     windows and IMDCT twiddles -- everything NVorbis builds in StreamDecoder.LoadBooks
//     (StreamDecoder.cs:226-289), Mode.Init (Mode.cs:24-67) and MdctImpl..ctor (Mdct.cs:30-63);
//   * the FRAME BATCH (per look-ahead batch of audio packets): what the bit-consuming half of
//     Mapping.DecodePacket (Mapping.cs:95-134) produced for each packet -- floor posts / LSP
//     coefficients, residue classifications + VQ entry numbers -- plus the integer frame geometry of
//     Mode.GetPacketInfo (Mode.cs:119-151) and StreamDecoder.ReadNextPacket (StreamDecoder.cs:417-463).
// No float ever travels host->device per packet except Floor0's amplitude/LSP values.
#pragma once
#include <stdint.h>

#define NVH_MAX_POSTS 64        // Floor1.Data.Posts = new int[64] (Floor1.cs:12)
#define NVH_LINK_NONE 0x7FFF     // NvhDevBatch::op_link: no later-stage op for this partition/channel
#define NVH_MAX_STAGES 8        // cascade is 8 bits wide (Residue0.cs:46-58)
#define NVH_MAX_CLASSES 64      // 6 bits + 1 (Residue0.cs:41)
#define NVH_ENTRY_SKIP 0xFFFFu  // entry-stream sentinel: "no vector was added here"

// ---- setup -------------------------------------------------------------------------------------

struct NvhDevBook {      // Codebook lookup table (Codebook.cs:222-283, indexer :322)
  uint32_t tab_off;      // float offset into the VQ pool; 0xFFFFFFFF for map type 0
  uint32_t entries;
  uint32_t dim;
  uint32_t dim_magic;    // ceil(2^32 / dim): x / dim == __umulhi(x, dim_magic) for x * dim < 2^32 (0 when dim <= 1)
  // Lattice books (Codebook.cs:242-260, map type 1 without sequence_p): component i of entry e is one of only
  // `lat_values` distinct floats, selected by digit (e / lat_values^i) % lat_values.  lat_off points into the
  // lattice pool: [lat_values floats][dim reciprocal magics of lat_values^i].  lat_values == 0: use the table.
  uint32_t lat_values;
  uint32_t lat_magic;    // ceil(2^32 / lat_values)
  uint32_t lat_off;
  uint32_t dim_magic16;  // ceil(2^16 / dim): i / dim == (i * dim_magic16) >> 16 for i < 4096, dim <= 16 (pair records)
};

struct NvhDevFloor1 {    // Floor1.cs:21-25, :93-133
  int32_t x_count;       // number of posts (<= NVH_MAX_POSTS on the decode path)
  int32_t multiplier;    // header field + 1 (Floor1.cs:74)
  int32_t range;
  int32_t levels;        // number of dependency levels of the unwrap (host-derived, see host_setup.cpp)
  uint16_t x_list[NVH_MAX_POSTS];
  uint8_t l_neigh[NVH_MAX_POSTS];
  uint8_t h_neigh[NVH_MAX_POSTS];
  uint8_t sort_idx[NVH_MAX_POSTS];
  uint8_t level[NVH_MAX_POSTS];  // post i can be unwrapped once all posts of lower level are final
  // derived per-post constants of the unwrap (one load level instead of two on the device)
  uint16_t x_lo[NVH_MAX_POSTS];      // x_list[l_neigh[i]]
  uint16_t x_hi[NVH_MAX_POSTS];      // x_list[h_neigh[i]]
  uint16_t x_sorted[NVH_MAX_POSTS];  // x_list[sort_idx[i]]
  uint32_t adx_magic[NVH_MAX_POSTS]; // floor((2^32 - 1) / (x_hi - x_lo)): quotient estimate, one fix-up step (i >= 2)
};

struct NvhDevFloor0 {    // Floor0.cs:22-26
  int32_t order;
  int32_t amp_ofs;
  int32_t bark_map_size;
  int32_t pad;
  uint32_t bark_off[2];  // int pool offsets of the Bark maps for block0 / block1 (n/2+1 ints each)
  uint32_t wmap_off[2];  // float pool offsets of the wdel maps for block0 / block1 (n/2 floats each)
};

struct NvhDevFloor {
  int32_t type;          // 0 / 1
  int32_t pad;
  NvhDevFloor0 f0;
  NvhDevFloor1 f1;
};

struct NvhDevResidue {   // Residue0.cs:21-33
  int32_t type;          // 0, 1, 2
  int32_t begin, end, partition_size;
  int32_t classifications;
  int32_t channels;      // channels seen by the base decode loop (1 for type 2, Residue2.cs:13)
  int32_t real_channels; // Residue2._channels
  int32_t sequential;    // 1 => partitions of a stage may alias (quirk B-1): apply ops in order
  uint32_t psize_magic;  // ceil(2^32 / partition_size)
  uint32_t rch_magic;    // ceil(2^32 / real_channels) (0 when real_channels == 1)
  int32_t fast;          // 1 => the reciprocal-multiply index path is exact for every index of this residue
  int32_t pair_path;     // 1 => fast, not sequential, every book a lattice book of even dimension: two bins per lane
  uint32_t hp_magic;     // ceil(2^32 / (partition_size / 2)) for the pair path, 0 when partition_size / 2 <= 1
  int32_t alias_b1;      // 1 => sequential only because of quirk B-1 (Residue2 partitions sharing a bin), lattice books of even
                         // dimension dividing the partition: the slab kernels walk it bin by bin (kernels_synth.hip: residue_walk_bins)
  int32_t pad[2];
};

struct NvhDevMapping {   // Mapping.cs:9-14
  int32_t coupling_steps;
  uint32_t coupling_off; // offset into the coupling pool: pairs (magnitude, angle) as uint8
};

// ---- per-batch frame descriptors -----------------------------------------------------------------

struct NvhResOp {        // one (stage, partition, channel) vector write (Residue0.cs:157-170)
  uint32_t ent_off;      // offset of its entries in the batch entry stream (uint16 each)
  uint16_t partition;
  uint8_t channel;       // channel index inside the residue (always 0 for type 2)
  uint8_t book;          // codebook index
};

struct NvhResPass {      // one IResidue.Decode call (Mapping.cs:133): ops grouped by stage
  int32_t residue;       // residue index
  uint32_t op_begin[NVH_MAX_STAGES + 1];  // ops of stage s are [op_begin[s], op_begin[s+1])
};

struct NvhChan {         // one channel of one frame ("ch-frame")
  uint8_t exec;          // IFloorData.ExecuteChannel after ForceEnergy / ForceNoEnergy (Mapping.cs:104-131)
  uint8_t floor;         // floor index
  uint8_t post_count;    // Floor1: Data.PostCount; Floor0: 1 if Amp > 0
  uint8_t ov_exec;       // exec flag of this channel in the frame the overlapped tail comes from
  uint32_t data_off;     // Floor1: offset of raw posts (uint16) in the post pool;
                         // Floor0: offset of coeff[order+1] (float) in the coeff pool
  float amp;             // Floor0 Data.Amp
};

struct NvhFrame {
  int32_t n;             // block size of the packet's mode
  int32_t mapping;       // mapping index
  uint32_t window_off;   // float offset of the window (n floats) in the window pool
  int32_t mdct_slot;     // 0 = block0 tables, 1 = block1 tables
  // geometry (Mode.cs:102-151, StreamDecoder.cs:417-463)
  int32_t start, valid, total;  // as returned by Mode.Decode (valid already EOS-trimmed)
  int32_t emit_start;    // first index of the block that is emitted
  int32_t emit_count;    // samples per channel this frame contributes to the PCM
  int32_t ov_frame;      // frame whose tail overlaps into this one: index in batch, -1 none, -2 carried tail
  int32_t ov_src;        // first index of that tail in the previous block
  int32_t ov_len;        // number of overlapped samples (added at [start, start+ov_len))
  int64_t out_pos;       // per-channel sample position of the first emitted sample in the batch PCM
  uint32_t chan_off;     // first NvhChan of this frame
  uint32_t pass_begin, pass_end;  // residue passes of this frame (one per submap)
  uint32_t op_begin, op_count;    // this frame's slice of the op list (contiguous, stage-major)
  uint32_t ent_begin, ent_count;  // this frame's slice of the entry stream (contiguous)
  int32_t ov_n;                   // block size of the frame the overlapped tail comes from (0 if none)
  uint32_t ov_window_off;         // window of that frame (its tail is stored un-windowed in the compact layout)
  uint32_t exec_mask;             // bit c = channel c executes (first 32 channels; mirrors NvhChan::exec)
  uint32_t ov_exec_mask;          // same for the overlap source frame (mirrors NvhChan::ov_exec)
  uint32_t emit_flags;            // NVH_EMIT_*: who overlap-adds this frame's PCM (set by the host at upload; 0 = k_ola_compact)
};

// In-kernel overlap-add of the slab synthesis kernel (kernels_synth.hip, "paired emission"): even frames of a batch are
// synthesised in a second launch, behind the odd ones, and emit the PCM of the steady-state overlaps they take part in --
// their own first half over frame f - 1's second half (SELF), and their second half under frame f + 1's first half (NEXT) --
// from registers and the odd frames' work planes, so that neither their own plane nor a k_ola_compact pass over those samples
// is needed.  DONE marks every frame whose PCM comes out of k_synth: k_ola_compact skips its overlap-add.
#define NVH_EMIT_SELF 1u
#define NVH_EMIT_NEXT 2u
#define NVH_EMIT_DONE 4u
// The two frames a steady-state batch would still leave to k_ola_compact: the batch's first frame, whose predecessor is the
// carried tail of the batch before (stored fully windowed: StreamDecoder's _prevPacketBuf) -- SELF_CARRY: the emitter adds the
// carried samples instead of a neighbour's quarter --, and the last decoded block, which becomes the next carried tail --
// CARRY_OUT: its workgroup writes the windowed block itself.  With both, a steady-state batch needs no k_ola_compact launch.
#define NVH_EMIT_SELF_CARRY 8u
#define NVH_EMIT_CARRY_OUT 16u

// ---- per-frame slabs of the slab synthesis kernel (kernels_synth.hip) ------------------------------------------------
//
// Everything k_synth needs about one frame, in its final LDS form, as ONE contiguous block at a fixed stride per batch,
// so that the workgroup fetches it with LDS-DMA in a single round trip (no frame record -> slices -> setup records chain
// of dependent loads, no staging copies, no unwrap in the kernel).  Written by the packet parsers (host_slab.cpp, kernels_parse.hip)
// (integer work only: Floor1.UnwrapPosts + the segment list of the sorted, flagged posts, Floor1.cs:196-297; the
// residue geometry of Residue0.cs:157-170 / Residue2.cs:23-47 resolved per vector write).  Sections, 16-byte aligned:
//   NvhSlabHdr | per channel: uint4 segment[nseg] (x | xend << 16, y, signed 32.32 step per bin), uint8 first_segment[n / 8] (one per four bins) |
//   uint32 heads[nheads] | uint2 rec[nrec] | uint16 entries[]
// rec = one (stage, partition, channel) vector write, 8 bytes (round 4; 16 before):
//   x: entry offset (frame relative) | ceil(2^16 / dim) << 16
//   y: lattice pool offset (12 bits) | lat_values << 12 (8) | dim << 20 (5) | channel << 25 (3) | stage << 28 (3) | more << 31
// (the reciprocal ceil(2^32 / lat_values) is the second of the book's power magics in the lattice pool: lat[lat_values + 1])
// laid out chain-major: the writes to one partition / channel through the cascade stages are consecutive records, in
// stage order, `more` set on all but the last; heads[k] = index of chain k's first record | the partition's first bin << 16.
#define NVH_SLAB_REC(ent_off, dm16, lat_off, lat_values, dim, channel, stage, more)                                              \
  ((uint32_t)(ent_off) | ((uint32_t)(dm16) << 16)),                                                                               \
      ((uint32_t)(lat_off) | ((uint32_t)(lat_values) << 12) | ((uint32_t)(dim) << 20) | ((uint32_t)(channel) << 25) |             \
       ((uint32_t)(stage) << 28) | ((more) ? 0x80000000u : 0u))
#define NVH_SLAB_MAX_LAT_OFF 0xFFFu  // lattice pool words a record can address
// Digit form of the entry section (round 5; NvhSlabHdr::rgeom bit 3): instead of uint16 entry numbers the section holds ONE BYTE
// PER VECTOR COMPONENT, record by record (a record's run = the partition's components in entry order: byte j * dim + d is
// component d of the record's entry j), value = 4 * (entry / lat_values^d % lat_values) -- the byte offset of the component's
// float from the book's first word in the value pool (host_slab.h: SlabSetup::val_pool; it follows the lattice pool in the
// kernels' constants block) -- or 4 * lat_values, the book's +0.0f slot, for a vector that was never added (quirks B-14 / B-16).
// A record's x then holds the run's offset in the section in units of 2 bytes, its y the book's value-pool offset where the
// entry form holds the lattice-pool offset.  The base-lat_values digit peel (two exact reciprocal multiplies + a multiply and a
// subtract per digit pair, ~110 VALU instructions per cascade stage of a lane's eight components) is integer work on what the bit
// parser decoded: it belongs to the parser, and the kernels' walk is a byte read, an LDS read and an add per component.
#define NVH_SLAB_RGEOM_DIGITS 8u
#define NVH_SLAB_MAX_DIGIT 63u       // lat_values of a book the digit form takes: 4 * lat_values fits a byte
#define NVH_SLAB_SWEEP_COUPLES 1u  // stereo Residue2: the chain walk holds both channels of a bin and couples before its store
#define NVH_SLAB_MG1 2u            // the magnitude channel of the coupling step is channel 1
#define NVH_SLAB_COUPLE_PASS 4u    // inverse coupling as a pass of its own between the residue walk and the floor multiply
#define NVH_SLAB_FLOOR_FAULT 8u    // a curve value outside inverse_dB_table (quirk B-7): the kernel raises NVH_DEVERR_FLOOR1_Y
#define NVH_SLAB_MDCT_SLOT 16u     // block1 tables (else block0)
#define NVH_SLAB_FUSE_FLOOR 32u    // the lane that finishes a chain multiplies its bins by the floor curve before its one store
#define NVH_SLAB_EMIT_SELF 64u     // NVH_EMIT_SELF of the frame (mono / stereo slabs only: chan[2..7] carry the parameters)
#define NVH_SLAB_EMIT_NEXT 128u    // NVH_EMIT_NEXT
// mono / stereo slabs: bits 6 and 7 of NvhSlabHdr::exec_mask (two channels need two bits)
#define NVH_SLABX_SELF_CARRY 0x40u // NVH_EMIT_SELF_CARRY (together with NVH_SLAB_EMIT_SELF)
#define NVH_SLABX_CARRY_OUT 0x80u  // NVH_EMIT_CARRY_OUT (chan[2] = window_off)
#define NVH_SLABX_DONE 0x20u       // NVH_EMIT_DONE: the frame's PCM comes out of the synthesis kernels -- its overlap with the frame before by
                                   // whoever holds SELF / NEXT, the samples only this block contributes to (a long block next to a short one:
                                   // the flat parts of its window, Mode.cs:102-117) by its own workgroup (kernels_synth.hip, frame groups)
// chan[5] of a mono / stereo slab with NVH_SLAB_EMIT_* / NVH_SLABX_DONE: the geometry of the overlaps in units of 64 samples
// (Mode.cs:102-151: every start / valid / block size of a stream with blocks >= 256 is a multiple of 64)
#define NVH_SLAB_GEO(prev_n, next_n, start, valid) \
  ((uint32_t)((prev_n) >> 6) | ((uint32_t)((next_n) >> 6) << 8) | ((uint32_t)((start) >> 6) << 16) | ((uint32_t)((valid) >> 6) << 24))
#define NVH_SLAB_HDR_VECS 4
#define NVH_SLAB_MAX_CH 8          // channels a slab describes (k_synth: 2, k_synth8: 8)
#define NVH_SLAB_MAX_COUPLE 4      // coupling steps of a pass of its own (3 + 3 bits each in NvhSlabHdr::coupling)
struct NvhSlabHdr {      // 64 bytes
  uint16_t n;            // block size of the packet's mode; 0 = pseudo-frame, nothing to compute
  uint8_t exec_mask;     // bit c: channel c executes (NvhChan::exec)
  uint8_t flags;         // NVH_SLAB_*
  uint16_t nheads, nrec;
  uint16_t off_heads, off_rec;  // section offsets in 16-byte units from the start of the slab
  uint16_t off_ent;
  uint16_t vecs;         // size of the slab in 16-byte units
  uint16_t lpc;          // lanes per chain = partition_size / group
  uint8_t rgeom;         // residue type (bits 0-2) | NVH_SLAB_RGEOM_DIGITS | real channels << 4
  uint8_t group;         // consecutive vector components one lane owns through all cascade stages: 8 (stereo / per-channel
                         // residues, partition_size % 8 == 0), 2 * channels (Residue2 over more than two channels), else 2
  uint32_t lpc_magic;    // ceil(2^32 / lpc), 0 when lpc <= 1
  uint32_t frame;        // the frame this slab belongs to: slabs are laid out in launch order (costliest frames first), not in frame order
  uint32_t coupling;     // NVH_SLAB_COUPLE_PASS: step count | (magnitude | angle << 3) << (4 + 6 k) for step k (Mapping.cs:137-182)
  uint32_t chan[NVH_SLAB_MAX_CH];  // per channel: floor mode (0 none, 1 curve, 2 clear; Floor1.cs:218-221) | nseg << 8 | off_seg << 16
                                   // NVH_SLAB_EMIT_* (at most two channels): chan[2..7] = window_off, ov_window_off, the next
                                   // frame's window_off, NVH_SLAB_GEO(block size of the frame before, of the frame behind, this
                                   // frame's start and valid), out_pos, the next frame's out_pos (the next frame's ov_window_off
                                   // is this frame's window_off)
};

