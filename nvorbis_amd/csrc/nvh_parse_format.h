// nvh_parse_format.h -- device-side tables for the GPU packet parser (kernels_parse.hip): the bit-consuming half
// of the decoder.  Plain-old-data mirrors of the host structures in host_setup.h,
// built once per setup by nvh_api.hip:upload_parse_tables.
//
// Reference behaviour these tables serve (file:line under /root/reference/NVorbis/):
//   Huffman.cs:15-76 (prefix table + overflow list), Codebook.cs:294-320 (DecodeScalar),
//   Floor1.cs:135-184 (Unpack), Residue0.cs:119-201 (Decode), Mapping.cs:95-134 (DecodePacket, bit half).
#pragma once
#include <stdint.h>

#include "nvh_format.h"

#define NVH_PARSE_MAX_CH 8        // channels the GPU parser handles (per-channel state lives in register bit masks)
#define NVH_PARSE_MAX_SUBMAPS 16  // Mapping.cs:45
#define NVH_PARSE_MAX_COUPLING 16

// Huffman decode tables of one codebook.
//   prefix[i] (1 << prefix_bits entries): (value << 8) | 0x80 | length for a code of at most prefix_bits bits, else
//     (group begin << 8) | group count: the longer codes that start with these prefix_bits (count 0x7F: scan all)
//   overflow: ovf_count nodes in the reference's (length, bits) order, then the same nodes grouped by prefix slot
struct NvhPBook {
  uint32_t prefix_off;    // into the uint32 prefix pool
  uint32_t ovf_off;       // into the NvhPOverflow pool
  uint32_t ovf_count;
  uint32_t entries;
  uint16_t dims;
  uint8_t prefix_bits, max_bits;
  uint8_t has_tree, has_overflow;  // Codebook.cs:294-320: `_prefixList != null`, `_overflowList != null`
  uint8_t pad[2];
  uint32_t lds_off;       // word offset of this book's prefix table inside the LDS image, 0xFFFFFFFF: read it from global memory
  // what a slab record says about the book (nvh_format.h: NVH_SLAB_REC; NvhDevBook has the same fields for the host writer)
  uint32_t slab_lat;      // lattice pool offset | lat_values << 16
  uint32_t slab_dm16;     // ceil(2^16 / dims)
  uint32_t dim_magic;     // ceil(2^32 / dims), 0 for dims <= 1: partition_size / dims without a division (nvh_setup.hip checks the range)
  uint32_t ovf_lds;       // word offset of the book's GROUPED overflow nodes inside the LDS image, two words each (bits, value << 8 |
                          // length; the mask is (1 << length) - 1), 0xFFFFFFFF: scan them in global memory
  uint32_t sub_dir;       // word offset of the book's directory inside the second-level image (NvhDevParse::sub_image; nvh_setup.hip),
                          // 0xFFFFFFFF: none -- a long code is found by scanning its group
};

struct NvhPOverflow {
  uint32_t bits, mask;
  uint32_t value;
  uint32_t length;
};

struct NvhPFloor1 {  // Floor1.cs:21-25 as Unpack uses it
  int32_t type;      // 1; anything else makes the stream ineligible for the GPU parser
  int32_t partition_count;
  int32_t y_bits;
  int32_t pad;
  uint8_t partition_class[32];
  uint8_t class_dims[16];
  uint8_t class_sub_bits[16];
  int16_t class_master[16];
  int16_t sub_book[16][8];
};

struct NvhPResidue {  // Residue0.cs:21-33
  int32_t type, begin, end, partition_size;
  int32_t classifications, class_book, channels, real_channels;
  int32_t max_stages, partvals, class_dims;
  int32_t alias_b1;         // quirk B-1 on its own (NvhDevResidue::alias_b1): the slab carries the partition table of the bin walk
  uint32_t decode_map_off;  // into the int pool: partvals * class_dims class numbers
  uint32_t rch_magic;       // ceil(2^32 / real_channels), 0 for one channel
  uint32_t general;         // its frames take the general bin walk (neither the pair path nor B-1 on its own; host_slab.h: residue_general)
  uint32_t decode_map_lds;  // word offset of the same class numbers inside the LDS image, 0xFFFFFFFF: read them from the int pool
  uint32_t vis_lds;         // word offset of the visit descriptors inside the LDS image (NVH_PVIS_*), 0xFFFFFFFF: none
  uint32_t span_max;        // components a partition's longest vector write covers (host_slab.h: residue_max_span): the bin walk's `cover`
  uint8_t cascade[NVH_MAX_CLASSES];
  int16_t books[NVH_MAX_CLASSES][NVH_MAX_STAGES];
  uint8_t book_mask[NVH_MAX_CLASSES];  // per class: the cascade stages that have a book (a chain of the slab has one record per set bit)
};

// Visit descriptors of a residue (k_parse_slab_u): four words per (class, cascade stage) -- everything a visit of the walk
// (Residue0.cs:157-170: one partition of one channel in one stage) needs to know once the partition's class is known, so that
// it is one 16-byte LDS read away from decoding: no cascade test, no book number, no 44-byte book record, no divisions.
//   x: NVH_PVIS_NONE (the class has no book in this stage: nothing to do), NVH_PVIS_SLOW (a book the fast form does not take: a
//      Residue0, a table in global memory, ...: the general code decides), else the book's prefix table in the LDS image (word
//      offset, 24 bits) | prefix_bits << 24
//   y: entries per partition | dims << 16 | rank << 24 (records of the chain in front of this stage's: popcount of the lower stages
//      that have a book)
//   z: the record's x word without the entry offset (ceil(2^16 / dims) << 16) | the book's number
//   w: the record's y word without the channel (nvh_format.h: NVH_SLAB_REC: pool offset, lat_values, dims, stage, more)
#define NVH_PVIS_NONE 0xFFFFFFFFu
#define NVH_PVIS_SLOW 0xFFFFFFFEu

struct NvhPMapping {  // Mapping.cs:16-78
  int32_t submaps, coupling_steps;
  uint8_t submap_floor[NVH_PARSE_MAX_SUBMAPS], submap_residue[NVH_PARSE_MAX_SUBMAPS];
  uint8_t chan_floor[NVH_PARSE_MAX_CH], chan_residue[NVH_PARSE_MAX_CH];
  uint8_t coupling_mag[NVH_PARSE_MAX_COUPLING], coupling_ang[NVH_PARSE_MAX_COUPLING];
};

struct NvhDevParse {
  int32_t channels, block1;
  int32_t cap_pass, cap_ops, cap_ent;  // per-frame slab capacities (worst case of the setup)
  int32_t cap_parts;                   // per-frame scratch: partitions * channels of the largest residue
  const NvhPBook* books;
  const uint32_t* prefix;
  const NvhPOverflow* overflow;
  const NvhPFloor1* floors;
  const NvhPResidue* residues;
  const NvhPMapping* mappings;
  const int32_t* ipool;
  const uint32_t* lds_image;  // prefix tables of the hottest books (residue VQ books first), copied into LDS by every workgroup
  int32_t lds_words;
  // books, floors, residues and mappings are contiguous in the arena (in that order): `meta_words` words from `books`
  // are copied into LDS too, the three offsets locate the other arrays inside that copy
  int32_t meta_words;
  int32_t meta_floors_off, meta_residues_off, meta_mappings_off;  // byte offsets from `books`
  // slab mode for streams of the general bin walk (several residue passes per frame, Residue0, odd dimensions, ...): the frame's
  // slab carries a group list (nvh_format.h: NvhSlabHdr::group == 1); the walk's scratch then keeps one chain-start row PER PASS
  int32_t slab_general;
  int32_t row_words;      // ints of per-packet scratch: the class-word row + the chain-start rows (kernels_parse.hip: parse_body)
  int32_t dm_in_lds;      // every residue's class decode map and visit descriptors have their copy in the LDS image (k_parse_slab_u
                          // reads nothing else)
  // slab mode (the parser writes the synthesis kernels' slabs itself): the synthesis setup's floor records and reciprocal table,
  // the slab stride (the setup's worst case, 16-byte units; 0: this setup's batches take the descriptor kernels)
  const NvhDevFloor* dfloors;
  const uint32_t* recip;
  int32_t slab_stride_vecs, max_posts;
  // second-level tables of the long codes (k_parse_slab_f): directories + tables, NvhPBook::sub_dir locates a book's directory
  const uint32_t* sub_image;
  int32_t sub_words, pad_sub;
};

// One packet's location for k_parse (its frame record carries the geometry).
struct NvhPacketRef {
  uint32_t byte_off;   // 4-byte aligned offset into the packet pool; the packet is zero-padded to a word boundary
  uint32_t bit_len;    // packet length in bits
  uint32_t bit_pos;    // bits the host already consumed (packet type, mode number, window flags)
  uint32_t pad;
};

// Batch-level results of k_parse, read back before the synthesis kernels are sized.
struct NvhParseResult {
  int32_t max_ops, max_ent, max_pass;
  int32_t err_frame;   // first frame (lowest index) whose packet would have made the reference throw, or 0x7FFFFFFF
  int32_t err_code;    // its NVH_ERR_* code
  int32_t links_ok;
  int32_t emit_ok;     // 1: every paired-emission candidate the host marked stands (all channels execute in the frames involved)
  int32_t max_vecs;    // slab mode: the batch's largest slab in 16-byte units (sizes the LDS slab area of the synthesis kernels)
};
