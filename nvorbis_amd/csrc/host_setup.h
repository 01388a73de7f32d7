// host_setup.h -- host-side model of a Vorbis stream setup (the three header packets).
//
// This is product code (not the oracle): it parses the identification and setup headers the way
// NVorbis does at construction time (StreamDecoder.cs:179-289, Factory.cs:22-58) and builds every
// table the GPU path needs.  All double-precision transcendental work happens here, on the host,
// with the reference's mixed float/double expression shapes.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/nvorbis_hip.h"
#include "host_bits.h"
#include "nvh_format.h"

namespace nvh {

// status codes: NVH_OK / NVH_ERR_* macros of the public C ABI header

struct HuffNode {
  int value = 0, length = 0, bits = 0, mask = 0;
  bool present = false;
};

// Codebook.cs + Huffman.cs
struct Codebook {
  int dimensions = 0, entries = 0, map_type = 0;
  std::vector<int> lengths;
  std::vector<float> lookup;      // entries * dimensions (map type != 0)
  int lattice_values = 0;         // map type 1 without sequence_p: number of distinct component values, else 0
  std::vector<float> lattice;     // those values: lookup[e*dim+i] == lattice[(e / lattice_values^i) % lattice_values]
  std::vector<HuffNode> prefix;   // 1 << prefix_bits
  // the same table in the packet parsers' compact form -- (value << 8) | 0x80 | length for a code that the prefix resolves, 0
  // else (kernels_parse.hip reads the same words from LDS) -- for the host parser's vector loop (host_parse.cpp)
  std::vector<uint32_t> fast;
  std::vector<HuffNode> overflow;
  // the overflow list regrouped by the first prefix_bits bits of each code (same relative order inside a group): a
  // code can only match a peek whose low bits select its group, so the reference's first-match scan over the whole
  // list (Codebook.cs:306-318) and a scan over the group return the same node.  slot_group[slot] = begin << 8 | count.
  std::vector<HuffNode> overflow_grouped;
  std::vector<uint32_t> slot_group;
  bool has_overflow = false;      // C# `_overflowList != null`
  bool has_tree = false;
  int prefix_bits = 0, max_bits = 0;

  int init(BitReader& p);
  // Codebook.cs:294-320.  -1 = no symbol; -2 = the reference would fault (null list)
  int decode_scalar(BitReader& p) const;
};

struct Floor1 {
  int partition_count = 0;
  int partition_class[32] = {0};
  int class_count = 0;
  int class_dimensions[16] = {0}, class_subclasses[16] = {0}, class_masterbook[16] = {0};
  int subclass_book[16][8];
  int multiplier = 0, range = 0, y_bits = 0;
  std::vector<int> x_list, l_neigh, h_neigh, sort_idx;
};

struct Floor0 {
  int order = 0, rate = 0, bark_map_size = 0, amp_bits = 0, amp_ofs = 0, amp_div = 0;
  int book_bits = 0;
  std::vector<int> books;
  std::vector<int> bark_map[2];   // [0] block0, [1] block1
  std::vector<float> w_map[2];
};

struct Floor {
  int type = 1;
  Floor0 f0;
  Floor1 f1;
};

struct Residue {
  int type = 0;
  int channels = 0;        // what the base decode loop iterates (1 for residue 2)
  int real_channels = 0;
  int begin = 0, end = 0, partition_size = 0, classifications = 0, max_stages = 0;
  int class_book = 0;
  int cascade[NVH_MAX_CLASSES] = {0};
  int books[NVH_MAX_CLASSES][NVH_MAX_STAGES];
  int partvals = 0;
  std::vector<int> decode_map;   // partvals * classbook.dimensions
};

struct Mapping {
  std::vector<int> coupling_angle, coupling_magnitude;
  std::vector<int> submap_floor, submap_residue;
  std::vector<int> channel_floor, channel_residue;
};

struct Mode {
  bool block_flag = false;
  int block_size = 0;
  int mapping = 0;
  uint32_t window_off[4] = {0, 0, 0, 0};   // float offsets into Setup::windows
  int ov_start[4] = {0}, ov_valid[4] = {0}, ov_total[4] = {0};
};

struct MdctTables {                // Mdct.cs:30-63
  int n = 0;
  std::vector<float> a, b, c;
  std::vector<uint16_t> bitrev;
  // _a re-ordered for the wavefront IMDCT kernel (kernels_imdct.hip): for every radix pass, the twiddle pairs a
  // lane needs, laid out [pair component][set] so that the 64 lanes of a wave read consecutive floats.
  // Same float values as `a` (bit copies); empty for n < 256.
  std::vector<float> tw;
  size_t fin_off = 0;  // float offset inside tw of the output stage's gather-address table (build_mdct_tables)
};

struct Setup {
  int channels = 0, sample_rate = 0, block0 = 0, block1 = 0;
  int upper_bitrate = 0, nominal_bitrate = 0, lower_bitrate = 0;  // IStreamDecoder.UpperBitrate / NominalBitrate / LowerBitrate
  int mode_field_bits = 0;
  std::vector<Codebook> books;
  std::vector<Floor> floors;
  std::vector<Residue> residues;
  std::vector<Mapping> mappings;
  std::vector<Mode> modes;
  std::vector<float> windows;      // pool of all windows
  MdctTables mdct[2];              // [0] block0, [1] block1

  // StreamDecoder.LoadStreamHeader / LoadComments(signature only) / LoadBooks
  int parse_id(const uint8_t* pkt, int len);
  int parse_comment_sig(const uint8_t* pkt, int len);
  int parse_setup(const uint8_t* pkt, int len);
};

// helpers exposed for unit tests and the fine-grained C ABI
int ilog(int x);
uint32_t bit_reverse(uint32_t n, int bits);
float vorbis_float32(uint32_t bits);
void calc_window(int prev_block, int block, int next_block, float* out);
void calc_overlap(int prev_block, int block, int next_block, int* start, int* valid, int* total);
void build_mdct_tables(int n, MdctTables& t);

}  // namespace nvh
