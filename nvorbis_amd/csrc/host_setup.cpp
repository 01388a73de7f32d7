// host_setup.cpp -- header-time parsing and table construction (product host code).
//
// Reference behaviour followed (file:line under /root/reference/NVorbis/):
//   Utils.cs:5-59, Huffman.cs:15-86, Codebook.cs:59-292, Floor1.cs:30-133, Floor0.cs:28-96,
//   Residue0.cs:35-117, Residue2.cs:10-14, Mapping.cs:16-93, Mode.cs:24-117, Mdct.cs:30-63,
//   Factory.cs:22-58, StreamDecoder.cs:145-289.
// Must be compiled with -ffp-contract=off: the float expressions that build tables are rounded per
// operation, like RyuJIT's scalar SSE code.
#include "host_setup.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace nvh {

int ilog(int x) {  // Utils.cs:5-14
  int cnt = 0;
  while (x > 0) { ++cnt; x >>= 1; }
  return cnt;
}

uint32_t bit_reverse(uint32_t n, int bits) {  // Utils.cs:21-28
  n = ((n & 0xAAAAAAAAu) >> 1) | ((n & 0x55555555u) << 1);
  n = ((n & 0xCCCCCCCCu) >> 2) | ((n & 0x33333333u) << 2);
  n = ((n & 0xF0F0F0F0u) >> 4) | ((n & 0x0F0F0F0Fu) << 4);
  n = ((n & 0xFF00FF00u) >> 8) | ((n & 0x00FF00FFu) << 8);
  n = (n >> 16) | (n << 16);
  return n >> ((32 - bits) & 31);  // C# masks the shift count of a 32-bit operand
}

float vorbis_float32(uint32_t bits) {  // Utils.cs:45-59
  int32_t sign = ((int32_t)bits) >> 31;
  double exponent = (double)((int32_t)((bits & 0x7fe00000u) >> 21) - 788);
  int64_t m = ((int64_t)(bits & 0x1fffffu) ^ (int64_t)sign) + (int64_t)(sign & 1);
  float mantissa = (float)m;
  return mantissa * (float)std::pow(2.0, exponent);
}

// ------------------------------------------------------------------------------------------------
// Codebook
// ------------------------------------------------------------------------------------------------

// Huffman.cs:15-76
static void generate_table(Codebook& cb, const int* values, const int* length_list, const int* code_list, int n) {
  const int kMaxTableBits = 10;  // Huffman.cs:9
  std::vector<HuffNode> list((size_t)n);
  int max_len = 0;
  for (int i = 0; i < n; i++) {
    list[i].value = values ? values[i] : i;
    list[i].length = length_list[i] <= 0 ? 99999 : length_list[i];
    list[i].bits = code_list[i];
    list[i].mask = (int)((1u << (length_list[i] & 31)) - 1u);
    list[i].present = true;
    if (length_list[i] > 0 && max_len < length_list[i]) max_len = length_list[i];
  }
  std::sort(list.begin(), list.end(), [](const HuffNode& x, const HuffNode& y) {
    if (x.length != y.length) return x.length < y.length;
    return x.bits < y.bits;
  });
  int table_bits = max_len > kMaxTableBits ? kMaxTableBits : max_len;
  cb.prefix.assign((size_t)1 << table_bits, HuffNode());
  cb.overflow.clear();
  cb.has_overflow = false;
  for (int i = 0; i < n && list[i].length < 99999; i++) {
    int item_bits = list[i].length;
    if (item_bits > table_bits) {
      cb.has_overflow = true;
      for (; i < n && list[i].length < 99999; i++) cb.overflow.push_back(list[i]);
    } else {
      int max_val = 1 << (table_bits - item_bits);
      for (int j = 0; j < max_val; j++) {
        int idx = (j << item_bits) | list[i].bits;
        if (idx >= 0 && idx < (int)cb.prefix.size()) cb.prefix[idx] = list[i];
      }
    }
  }
  cb.prefix_bits = table_bits;
  cb.has_tree = true;
  cb.fast.assign(cb.prefix.size(), 0u);
  for (size_t k = 0; k < cb.prefix.size(); k++) {
    const HuffNode& nd = cb.prefix[k];
    if (nd.present && nd.length >= 1 && nd.length <= table_bits && nd.value >= 0 && nd.value <= 0xFFFF)
      cb.fast[k] = ((uint32_t)nd.value << 8) | 0x80u | (uint32_t)nd.length;
  }
  // group the overflow list by prefix slot (stable)
  cb.overflow_grouped.clear();
  cb.slot_group.assign(cb.prefix.size(), 0u);
  if (cb.has_overflow) {
    const uint32_t slot_mask = (uint32_t)cb.prefix.size() - 1u;
    std::vector<uint32_t> count(cb.prefix.size(), 0u), begin(cb.prefix.size(), 0u);
    for (const HuffNode& nd : cb.overflow) count[(uint32_t)nd.bits & slot_mask]++;
    uint32_t run = 0;
    for (size_t k = 0; k < count.size(); k++) {
      begin[k] = run;
      run += count[k];
    }
    cb.overflow_grouped.resize(cb.overflow.size());
    std::vector<uint32_t> fill(begin);
    for (const HuffNode& nd : cb.overflow) cb.overflow_grouped[fill[(uint32_t)nd.bits & slot_mask]++] = nd;
    bool wide = false;  // a code of 32+ bits has a wrapped mask (Huffman.cs:29) and could match any slot: keep the plain scan
    for (const HuffNode& nd : cb.overflow) wide = wide || nd.length >= 32 || nd.length <= table_bits;
    for (size_t k = 0; k < count.size(); k++) cb.slot_group[k] = (begin[k] << 8) | ((count[k] < 0xFFu && !wide) ? count[k] : 0xFFu);
  }
}

// Codebook.cs:172-220.  1 ok, 0 over-subscribed, -1 runtime fault (32-bit lengths)
static int compute_codewords(bool sparse, int* codewords, int* codeword_lengths, const int* len, int n, int* values) {
  uint32_t available[33];
  std::memset(available, 0, sizeof available);
  int k, m = 0;
  for (k = 0; k < n; ++k)
    if (len[k] > 0) break;
  if (k == n) return 1;
  auto add_entry = [&](uint32_t code, int symbol, int count, int l) {
    if (sparse) {
      codewords[count] = (int)code;
      codeword_lengths[count] = l;
      values[count] = symbol;
    } else {
      codewords[symbol] = (int)code;
    }
  };
  add_entry(0u, k, m++, len[k]);
  if (len[k] > 31) return -1;
  for (int i = 1; i <= len[k]; ++i) available[i] = 1u << (32 - i);
  for (int i = k + 1; i < n; ++i) {
    int z = len[i];
    if (z <= 0) continue;
    if (z > 31) return -1;
    while (z > 0 && available[z] == 0) --z;
    if (z == 0) return 0;
    uint32_t res = available[z];
    available[z] = 0;
    add_entry(bit_reverse(res, 32), i, m++, len[i]);
    if (z != len[i])
      for (int y = len[i]; y > z; --y) available[y] = res + (1u << (32 - y));
  }
  return 1;
}

static int lookup1_values(int entries, int dimensions) {  // Codebook.cs:285-292
  int r = (int)std::floor(std::exp(std::log((double)entries) / dimensions));
  if (std::floor(std::pow((double)(r + 1), (double)dimensions)) <= entries) ++r;
  return r;
}

int Codebook::init(BitReader& p) {
  // Codebook.cs:59-74
  if (p.read(24) != 0x564342ull) return NVH_ERR_INVALID_DATA;
  dimensions = (int)p.read(16);
  entries = (int)p.read(24);
  lengths.assign((size_t)entries, 0);

  // InitTree (Codebook.cs:76-170)
  bool sparse;
  int total = 0, max_len;
  if (p.read_bit()) {
    int len = (int)p.read(5) + 1;
    for (int i = 0; i < entries;) {
      // The reference's loop (Codebook.cs:84-97) has no exit when the packet runs out: every further read returns 0,
      // `i` stops advancing and `len` counts up for ever.  A hang is not behaviour to mirror across a C ABI that is
      // handed untrusted files: a truncated header, a length no code can have, or a run past the last entry is
      // reported as invalid data.
      int cnt = (int)p.read(ilog(entries - i));
      if (p.is_short || len > 32) return NVH_ERR_INVALID_DATA;
      if (cnt > entries - i) return NVH_ERR_RUNTIME;  // lengths[i++] past the array: IndexOutOfRangeException
      while (--cnt >= 0) lengths[i++] = len;
      ++len;
    }
    total = 0;
    sparse = false;
    max_len = len;
  } else {
    max_len = -1;
    sparse = p.read_bit();
    for (int i = 0; i < entries; i++) {
      if (!sparse || p.read_bit()) {
        lengths[i] = (int)p.read(5) + 1;
        ++total;
      } else {
        lengths[i] = -1;
      }
      if (lengths[i] > max_len) max_len = lengths[i];
    }
  }
  max_bits = max_len;
  if (max_len > -1) {
    std::vector<int> codeword_lengths, values, codewords;
    bool have_cwl = false;
    if (sparse && total >= (entries >> 2)) {
      codeword_lengths = lengths;
      have_cwl = true;
      sparse = false;
    }
    int sorted_count = sparse ? total : 0;
    if (!sparse) {
      codewords.assign((size_t)entries, 0);
    } else {
      codeword_lengths.assign((size_t)sorted_count, 0);
      codewords.assign((size_t)sorted_count, 0);
      values.assign((size_t)sorted_count, 0);
      have_cwl = true;
    }
    int rc = compute_codewords(sparse, codewords.data(), codeword_lengths.data(), lengths.data(), entries,
                               values.data());
    if (rc < 0) return NVH_ERR_RUNTIME;
    if (rc == 0) return NVH_ERR_INVALID_DATA;  // Codebook.cs:161
    generate_table(*this, sparse ? values.data() : nullptr, have_cwl ? codeword_lengths.data() : lengths.data(),
                   codewords.data(), (int)codewords.size());
  }

  // InitLookupTable (Codebook.cs:222-283)
  map_type = (int)p.read(4);
  if (map_type == 0) return NVH_OK;
  float min_value = vorbis_float32((uint32_t)p.read(32));
  float delta_value = vorbis_float32((uint32_t)p.read(32));
  int value_bits = (int)p.read(4) + 1;
  bool sequence_p = p.read_bit();
  int64_t table_len = (int64_t)entries * dimensions;
  // Documented limit of this build: a lookup table of more than 2^26 values (256 MB of floats;
  // the largest table of any shipped file has 52 488) is refused before anything is sized by it -- the header fields
  // are untrusted, entries * dimensions can reach 2^40.
  if (table_len > ((int64_t)1 << 26)) return NVH_ERR_UNSUPPORTED;
  if (p.is_short) return NVH_ERR_INVALID_DATA;  // truncated header: nothing below would be meaningful
  int lookup_value_count = (int)table_len;
  if (map_type == 1) {
    if (dimensions == 0 || entries == 0) return NVH_ERR_RUNTIME;
    lookup_value_count = lookup1_values(entries, dimensions);
    if (lookup_value_count <= 0) return NVH_ERR_RUNTIME;
  }
  std::vector<uint32_t> mult((size_t)std::max(lookup_value_count, 0));
  for (int i = 0; i < lookup_value_count; i++) mult[i] = (uint32_t)p.read(value_bits);
  lookup.assign((size_t)table_len, 0.0f);
  lattice_values = 0;
  lattice.clear();
  if (map_type == 1 && !sequence_p) {
    // every component is (float)mult[digit] * delta + min, widened to double, + 0.0, narrowed again: only
    // lookup_value_count distinct results, built here with the very expression of the table loop below
    lattice_values = lookup_value_count;
    for (int k = 0; k < lookup_value_count; k++) {
      float fv = (float)mult[k] * delta_value;
      fv = fv + min_value;
      double value = (double)fv + 0.0;
      lattice.push_back((float)value);
    }
  }
  if (map_type == 1) {
    for (int idx = 0; idx < entries; idx++) {
      double last = 0.0;
      int idx_div = 1;
      for (int i = 0; i < dimensions; i++) {
        if (idx_div == 0) return NVH_ERR_RUNTIME;
        int moff = (idx / idx_div) % lookup_value_count;
        float fv = (float)mult[moff] * delta_value;  // float * float
        fv = fv + min_value;                         // float + float
        double value = (double)fv + last;            // + double (Codebook.cs:255)
        lookup[(size_t)idx * dimensions + i] = (float)value;
        if (sequence_p) last = value;
        idx_div = (int)((uint32_t)idx_div * (uint32_t)lookup_value_count);
      }
    }
  } else {
    for (int idx = 0; idx < entries; idx++) {
      double last = 0.0;
      int moff = idx * dimensions;
      for (int i = 0; i < dimensions; i++) {
        float fv = (float)mult[moff] * delta_value;
        fv = fv + min_value;
        double value = (double)fv + last;            // Codebook.cs:272
        lookup[(size_t)idx * dimensions + i] = (float)value;
        if (sequence_p) last = value;
        ++moff;
      }
    }
  }
  return NVH_OK;
}

int Codebook::decode_scalar(BitReader& p) const {
  int got;
  int data = (int)p.peek(prefix_bits, &got);
  if (got == 0) return -1;
  if (!has_tree) return -2;
  const HuffNode& node = prefix[(size_t)data];
  if (node.present) {
    p.skip(node.length);
    return node.value;
  }
  const uint32_t group = has_overflow ? slot_group[(size_t)data] : 0u;
  data = (int)p.peek(max_bits, &got);
  if (!has_overflow) return -2;
  if ((group & 0xFFu) == 0xFFu) {  // oversized group: the plain scan
    for (const HuffNode& n : overflow) {
      if (n.bits == (data & n.mask)) {
        p.skip(n.length);
        return n.value;
      }
    }
    return -1;
  }
  const HuffNode* g = overflow_grouped.data() + (group >> 8);
  for (uint32_t k = 0; k < (group & 0xFFu); k++) {
    if (g[k].bits == (data & g[k].mask)) {
      p.skip(g[k].length);
      return g[k].value;
    }
  }
  return -1;
}

// ------------------------------------------------------------------------------------------------
// Floors
// ------------------------------------------------------------------------------------------------

static int floor1_init(Floor1& f, BitReader& p, int nbooks) {  // Floor1.cs:30-133
  static const int range_lookup[4] = {256, 128, 86, 64};
  static const int ybits_lookup[4] = {8, 7, 7, 6};
  int maximum_class = -1;
  f.partition_count = (int)p.read(5);
  for (int i = 0; i < f.partition_count; i++) {
    f.partition_class[i] = (int)p.read(4);
    maximum_class = std::max(maximum_class, f.partition_class[i]);
  }
  f.class_count = ++maximum_class;
  for (int i = 0; i < maximum_class; i++) {
    f.class_dimensions[i] = (int)p.read(3) + 1;
    f.class_subclasses[i] = (int)p.read(2);
    f.class_masterbook[i] = -1;
    if (f.class_subclasses[i] > 0) {
      f.class_masterbook[i] = (int)p.read(8);
      if (f.class_masterbook[i] >= nbooks) return NVH_ERR_RUNTIME;
    }
    for (int j = 0; j < 8; j++) f.subclass_book[i][j] = -1;
    for (int j = 0; j < (1 << f.class_subclasses[i]); j++) {
      int book_num = (int)p.read(8) - 1;
      if (book_num >= nbooks) return NVH_ERR_RUNTIME;
      f.subclass_book[i][j] = book_num;
    }
  }
  f.multiplier = (int)p.read(2);
  f.range = range_lookup[f.multiplier];
  f.y_bits = ybits_lookup[f.multiplier];
  ++f.multiplier;
  int range_bits = (int)p.read(4);
  f.x_list.clear();
  f.x_list.push_back(0);
  f.x_list.push_back(1 << range_bits);
  for (int i = 0; i < f.partition_count; i++) {
    int cls = f.partition_class[i];
    for (int j = 0; j < f.class_dimensions[cls]; j++) f.x_list.push_back((int)p.read(range_bits));
  }
  int cnt = (int)f.x_list.size();
  f.l_neigh.assign((size_t)cnt, 0);
  f.h_neigh.assign((size_t)cnt, 0);
  f.sort_idx.assign((size_t)cnt, 0);
  f.sort_idx[0] = 0;
  f.sort_idx[1] = 1;
  for (int i = 2; i < cnt; i++) {
    f.l_neigh[i] = 0;
    f.h_neigh[i] = 1;
    f.sort_idx[i] = i;
    for (int j = 2; j < i; j++) {
      int temp = f.x_list[j];
      if (temp < f.x_list[i]) {
        if (temp > f.x_list[f.l_neigh[i]]) f.l_neigh[i] = j;
      } else {
        if (temp < f.x_list[f.h_neigh[i]]) f.h_neigh[i] = j;
      }
    }
  }
  for (int i = 0; i < cnt - 1; i++) {
    for (int j = i + 1; j < cnt; j++) {
      if (f.x_list[i] == f.x_list[j]) return NVH_ERR_INVALID_DATA;
      if (f.x_list[f.sort_idx[i]] > f.x_list[f.sort_idx[j]]) std::swap(f.sort_idx[i], f.sort_idx[j]);
    }
  }
  return NVH_OK;
}

static float to_bark(double lsp) {  // Floor0.cs:81-84
  return (float)(13.1 * std::atan(0.00074 * lsp) + 2.24 * std::atan(0.0000000185 * lsp * lsp) + .0001 * lsp);
}

static int floor0_init(Floor0& f, BitReader& p, int block0, int block1, const std::vector<Codebook>& books) {
  // Floor0.cs:28-65
  f.order = (int)p.read(8);
  f.rate = (int)p.read(16);
  f.bark_map_size = (int)p.read(16);
  f.amp_bits = (int)p.read(6);
  f.amp_ofs = (int)p.read(8);
  int nb = (int)p.read(4) + 1;
  if (f.order < 1 || f.rate < 1 || f.bark_map_size < 1 || nb == 0) return NVH_ERR_INVALID_DATA;
  f.amp_div = (int)((1u << (f.amp_bits & 31)) - 1u);
  f.books.clear();
  for (int i = 0; i < nb; i++) {
    int num = (int)p.read(8);
    if (num < 0 || num >= (int)books.size()) return NVH_ERR_INVALID_DATA;
    if (books[num].map_type == 0 || books[num].dimensions < 1) return NVH_ERR_INVALID_DATA;
    f.books.push_back(num);
  }
  f.book_bits = ilog(nb);
  const int sizes[2] = {block0 / 2, block1 / 2};
  for (int w = 0; w < 2; w++) {
    int n = sizes[w];
    // SynthesizeBarkCurve (Floor0.cs:67-79)
    float scale = (float)f.bark_map_size / to_bark((double)(f.rate / 2));
    f.bark_map[w].assign((size_t)n + 1, 0);
    for (int i = 0; i < n - 1; i++) {
      float hz = (float)f.rate / 2.0f;
      hz = hz / (float)n;
      hz = hz * (float)i;
      float t = to_bark((double)hz) * scale;
      int v = (int)std::floor((double)t);
      f.bark_map[w][i] = std::min(f.bark_map_size - 1, v);
    }
    f.bark_map[w][n] = -1;
    // SynthesizeWDelMap (Floor0.cs:86-96)
    float wdel = (float)(3.14159265358979323846 / f.bark_map_size);
    f.w_map[w].assign((size_t)n, 0.0f);
    for (int i = 0; i < n; i++) {
      float arg = wdel * (float)i;
      f.w_map[w][i] = 2.0f * (float)std::cos((double)arg);
    }
  }
  return NVH_OK;
}

// ------------------------------------------------------------------------------------------------
// Residue / Mapping / Mode
// ------------------------------------------------------------------------------------------------

static int icount(int v) {
  int r = 0;
  while (v != 0) { r += v & 1; v = (int)((unsigned)v >> 1); }
  return r;
}

static int residue_init(Residue& r, int type, BitReader& p, int channels, const std::vector<Codebook>& books) {
  // Residue0.cs:35-117, Residue2.cs:10-14
  int nbooks = (int)books.size();
  r.type = type;
  r.real_channels = channels;
  r.begin = (int)p.read(24);
  r.end = (int)p.read(24);
  r.partition_size = (int)p.read(24) + 1;
  r.classifications = (int)p.read(6) + 1;
  r.class_book = (int)p.read(8);
  if (r.class_book >= nbooks) return NVH_ERR_RUNTIME;
  int acc = 0;
  for (int i = 0; i < r.classifications; i++) {
    int low_bits = (int)p.read(3);
    if (p.read_bit()) r.cascade[i] = ((int)p.read(5) << 3) | low_bits;
    else r.cascade[i] = low_bits;
    acc += icount(r.cascade[i]);
  }
  std::vector<int> book_nums((size_t)acc);
  for (int i = 0; i < acc; i++) {
    book_nums[i] = (int)p.read(8);
    if (book_nums[i] >= nbooks) return NVH_ERR_RUNTIME;
    if (books[book_nums[i]].map_type == 0) return NVH_ERR_INVALID_DATA;
  }
  int entries = books[r.class_book].entries;
  int dim = books[r.class_book].dimensions;
  int partvals = 1;
  while (dim > 0) {
    partvals *= r.classifications;
    if (partvals > entries) return NVH_ERR_INVALID_DATA;
    --dim;
  }
  acc = 0;
  int maxstage = 0;
  for (int j = 0; j < r.classifications; j++) {
    int stages = ilog(r.cascade[j]);
    for (int k = 0; k < NVH_MAX_STAGES; k++) r.books[j][k] = -1;
    if (stages > 0) {
      maxstage = std::max(maxstage, stages);
      for (int k = 0; k < stages; k++)
        if ((r.cascade[j] & (1 << k)) > 0) r.books[j][k] = book_nums[acc++];
    }
  }
  r.max_stages = maxstage;
  dim = books[r.class_book].dimensions;
  r.partvals = partvals;
  r.decode_map.assign((size_t)partvals * (size_t)std::max(dim, 1), 0);
  for (int j = 0; j < partvals; j++) {
    int val = j;
    int mult = partvals / r.classifications;
    for (int k = 0; k < dim; k++) {
      if (mult == 0) return NVH_ERR_RUNTIME;
      int deco = val / mult;
      val -= deco * mult;
      mult /= r.classifications;
      r.decode_map[(size_t)j * dim + k] = deco;
    }
  }
  r.channels = (type == 2) ? 1 : channels;
  return NVH_OK;
}

static int mapping_init(Mapping& m, BitReader& p, int channels, int nfloors, int nresidues) {  // Mapping.cs:16-93
  int submap_count = 1, coupling_steps = 0;
  if (p.read_bit()) submap_count += (int)p.read(4);
  if (p.read_bit()) coupling_steps = (int)p.read(8) + 1;
  int coupling_bits = ilog(channels - 1);
  for (int j = 0; j < coupling_steps; j++) {
    int magnitude = (int)p.read(coupling_bits);
    int angle = (int)p.read(coupling_bits);
    if (magnitude == angle || magnitude > channels - 1 || angle > channels - 1) return NVH_ERR_INVALID_DATA;
    m.coupling_angle.push_back(angle);
    m.coupling_magnitude.push_back(magnitude);
  }
  if (0 != p.read(2)) return NVH_ERR_INVALID_DATA;
  std::vector<int> mux((size_t)channels, 0);
  if (submap_count > 1) {
    for (int c = 0; c < channels; c++) {
      mux[c] = (int)p.read(4);
      if (mux[c] > submap_count) return NVH_ERR_INVALID_DATA;  // sic (Mapping.cs:53)
    }
  }
  for (int j = 0; j < submap_count; j++) {
    p.skip(8);
    int floor_num = (int)p.read(8);
    if (floor_num >= nfloors) return NVH_ERR_INVALID_DATA;
    int residue_num = (int)p.read(8);
    if (residue_num >= nresidues) return NVH_ERR_INVALID_DATA;
    m.submap_floor.push_back(floor_num);
    m.submap_residue.push_back(residue_num);
  }
  for (int c = 0; c < channels; c++) {
    if (mux[c] >= submap_count) return NVH_ERR_RUNTIME;
    m.channel_floor.push_back(m.submap_floor[mux[c]]);
    m.channel_residue.push_back(m.submap_residue[mux[c]]);
  }
  return NVH_OK;
}

void calc_window(int prev_block, int block, int next_block, float* array) {  // Mode.cs:69-100
  const float M_PI2 = 3.1415926539f / 2;  // Mode.cs:15
  int left = prev_block / 2, wnd = block, right = next_block / 2;
  int leftbegin = wnd / 4 - left / 2;
  int rightbegin = wnd - wnd / 4 - right / 2;
  for (int i = 0; i < block; i++) array[i] = 0.0f;
  for (int i = 0; i < left; i++) {
    float x = (float)std::sin((i + .5) / left * (double)M_PI2);
    x *= x;
    float y = x * M_PI2;
    array[leftbegin + i] = (float)std::sin((double)y);
  }
  for (int i = leftbegin + left; i < rightbegin; i++) array[i] = 1.0f;
  for (int i = 0; i < right; i++) {
    float x = (float)std::sin((right - i - .5) / right * (double)M_PI2);
    x *= x;
    float y = x * M_PI2;
    array[rightbegin + i] = (float)std::sin((double)y);
  }
}

void calc_overlap(int prev_block, int block, int next_block, int* start, int* valid, int* total) {  // Mode.cs:102-117
  int left_half = prev_block / 4, right_half = next_block / 4;
  *start = block / 4 - left_half;
  *total = block / 4 * 3 + right_half;
  *valid = *total - right_half * 2;
}

void build_mdct_tables(int n, MdctTables& t) {  // Mdct.cs:30-63
  const float M_PI_F = 3.14159265358979323846264f;
  int n2 = n >> 1, n4 = n2 >> 1, n8 = n4 >> 1;
  int ld = ilog(n) - 1;
  t.n = n;
  t.a.assign((size_t)n2, 0.0f);
  t.b.assign((size_t)n2, 0.0f);
  t.c.assign((size_t)n4, 0.0f);
  t.bitrev.assign((size_t)n8, 0);
  for (int k = 0, k2 = 0; k < n4; ++k, k2 += 2) {
    float arg_a = (float)(4 * k) * M_PI_F;
    arg_a = arg_a / (float)n;
    float arg_b = (float)(k2 + 1) * M_PI_F;
    arg_b = arg_b / (float)n;
    arg_b = arg_b / 2.0f;
    t.a[k2] = (float)std::cos((double)arg_a);
    t.a[k2 + 1] = (float)-std::sin((double)arg_a);
    t.b[k2] = (float)std::cos((double)arg_b) * .5f;
    t.b[k2 + 1] = (float)std::sin((double)arg_b) * .5f;
  }
  for (int k = 0, k2 = 0; k < n8; ++k, k2 += 2) {
    float arg_c = (float)(2 * (k2 + 1)) * M_PI_F;
    arg_c = arg_c / (float)n;
    t.c[k2] = (float)std::cos((double)arg_c);
    t.c[k2 + 1] = (float)-std::sin((double)arg_c);
  }
  for (int i = 0; i < n8; ++i) t.bitrev[i] = (uint16_t)(bit_reverse((uint32_t)i, ld - 3) << 2);

  // Lane-ordered twiddles.  The (ld-5) radix-2 stages with distances N/2 .. 8 (N = n/4 complex points) are run
  // in passes of up to three stages; pass (R, S) handles distances S<<(R-1) .. S on sets base + S*k, set index
  // s = blk*S + r.  The butterfly (c_lo, c_lo + D) uses _a[t], _a[t+1], t = (D-1-(c_lo mod D)) * (n2/D), and
  // c_lo mod D = r + S*kk with kk = k mod 2^st.  Pair index: stages from the largest distance down, kk ascending.
  t.tw.clear();
  if (n >= 256) {
    const int N = n >> 2;
    int remain = ld - 5;
    while (remain > 0) {
      const int R = remain >= 3 ? 3 : remain;
      const int S = 8 << (remain - R);
      const int nsets = N >> R;
      const int pairs = (1 << R) - 1;
      size_t base = t.tw.size();
      t.tw.resize(base + (size_t)2 * pairs * nsets);
      for (int s = 0; s < nsets; ++s) {
        const int r = s & (S - 1);
        int pi = 0;
        for (int st = R - 1; st >= 0; --st) {
          const int D = S << st;
          for (int kk = 0; kk < (1 << st); ++kk, ++pi) {
            const int m = D - 1 - (r + S * kk);
            const int ti = m * (n2 / D);
            t.tw[base + (size_t)(2 * pi) * nsets + s] = t.a[(size_t)ti];
            t.tw[base + (size_t)(2 * pi + 1) * nsets + s] = t.a[(size_t)ti + 1];
          }
        }
      }
      remain -= R;
    }
    // The output stage's gather addresses (imdct_wave.h: steps 4-6 fused with 7 and 8).  Pair index p reads the four float2 slots
    // u[BR[2i]], u[BR[2i+1]] (i = 2p + h) and u[BR[2i']+2], u[BR[2i'+1]+2] (i' = n/16 - 1 - i) for h = 0, 1 from the rotated
    // layout the last pass leaves: eight slot numbers that depend on the lane alone, ~100 integer instructions per lane to
    // derive -- one 16-byte table entry per pair instead, behind the pass twiddles (16-byte aligned): uint16 x 8 =
    // e0, e1, g0, g1 of h = 0, then of h = 1.
    while (t.tw.size() & 3u) t.tw.push_back(0.0f);
    t.fin_off = t.tw.size();
    auto phys_rot = [](int c) {
      const int b = c >> 3, p = c & 7;
      return ((b + (b >> 3)) << 3) + ((((p >> 1) + (b >> 2)) & 3) << 1) + (p & 1);
    };
    const int npairs = n >> 5;
    t.tw.resize(t.fin_off + (size_t)4 * npairs);
    uint16_t* f = reinterpret_cast<uint16_t*>(t.tw.data() + t.fin_off);
    for (int p = 0; p < npairs; ++p)
      for (int h = 0; h < 2; ++h) {
        const int i = 2 * p + h, ir = (n >> 4) - 1 - i;
        const int kE0 = (int)bit_reverse((uint32_t)(2 * i), ld - 3) << 2, kE1 = (int)bit_reverse((uint32_t)(2 * i + 1), ld - 3) << 2;
        const int kD0 = (int)bit_reverse((uint32_t)(2 * ir), ld - 3) << 2, kD1 = (int)bit_reverse((uint32_t)(2 * ir + 1), ld - 3) << 2;
        f[8 * p + 4 * h + 0] = (uint16_t)phys_rot(kE0 >> 1);
        f[8 * p + 4 * h + 1] = (uint16_t)phys_rot(kE1 >> 1);
        f[8 * p + 4 * h + 2] = (uint16_t)phys_rot((kD0 >> 1) + 1);
        f[8 * p + 4 * h + 3] = (uint16_t)phys_rot((kD1 >> 1) + 1);
      }
  }
}

static int mode_init(Mode& m, BitReader& p, Setup& s) {  // Mode.cs:24-67
  m.block_flag = p.read_bit();
  if (0 != p.read(32)) return NVH_ERR_INVALID_DATA;
  m.mapping = (int)p.read(8);
  if (m.mapping >= (int)s.mappings.size()) return NVH_ERR_INVALID_DATA;
  if (m.block_flag) {
    static const int prevsel[4] = {0, 1, 0, 1}, nextsel[4] = {0, 0, 1, 1};
    m.block_size = s.block1;
    for (int i = 0; i < 4; i++) {
      int pb = prevsel[i] ? s.block1 : s.block0, nb = nextsel[i] ? s.block1 : s.block0;
      m.window_off[i] = (uint32_t)s.windows.size();
      s.windows.resize(s.windows.size() + (size_t)s.block1);
      calc_window(pb, s.block1, nb, s.windows.data() + m.window_off[i]);
      calc_overlap(pb, s.block1, nb, &m.ov_start[i], &m.ov_valid[i], &m.ov_total[i]);
    }
  } else {
    m.block_size = s.block0;
    m.window_off[0] = (uint32_t)s.windows.size();
    s.windows.resize(s.windows.size() + (size_t)s.block0);
    calc_window(s.block0, s.block0, s.block0, s.windows.data() + m.window_off[0]);
  }
  return NVH_OK;
}

// ------------------------------------------------------------------------------------------------
// Header packets
// ------------------------------------------------------------------------------------------------

static bool validate_header(BitReader& p, const uint8_t* sig, int n) {
  for (int i = 0; i < n; i++)
    if (sig[i] != p.read(8)) return false;
  return true;
}

int Setup::parse_id(const uint8_t* pkt, int len) {  // StreamDecoder.cs:179-204
  static const uint8_t sig[11] = {0x01, 0x76, 0x6f, 0x72, 0x62, 0x69, 0x73, 0, 0, 0, 0};
  BitReader p(pkt, len);
  if (!validate_header(p, sig, 11)) return NVH_ERR_NOT_VORBIS;
  channels = (int)(uint8_t)p.read(8);
  sample_rate = (int)p.read(32);
  upper_bitrate = (int)p.read(32);    // StreamDecoder.cs:191-193: (int)packet.ReadBits(32)
  nominal_bitrate = (int)p.read(32);
  lower_bitrate = (int)p.read(32);
  block0 = 1 << (int)p.read(4);
  block1 = 1 << (int)p.read(4);
  if (nominal_bitrate == 0 && upper_bitrate > 0 && lower_bitrate > 0)  // StreamDecoder.cs:196-199
    nominal_bitrate = (int)((uint32_t)upper_bitrate + (uint32_t)lower_bitrate) / 2;  // the managed int sum wraps
  if (channels < 1) return NVH_ERR_RUNTIME;  // count % _channels divides by zero in the reference
  return NVH_OK;
}

int Setup::parse_comment_sig(const uint8_t* pkt, int len) {  // StreamDecoder.cs:206-224 (signature only)
  static const uint8_t sig[7] = {0x03, 0x76, 0x6f, 0x72, 0x62, 0x69, 0x73};
  BitReader p(pkt, len);
  return validate_header(p, sig, 7) ? NVH_OK : NVH_ERR_NOT_VORBIS;
}

int Setup::parse_setup(const uint8_t* pkt, int len) {  // StreamDecoder.cs:226-289
  static const uint8_t sig[7] = {0x05, 0x76, 0x6f, 0x72, 0x62, 0x69, 0x73};
  BitReader p(pkt, len);
  if (!validate_header(p, sig, 7)) return NVH_ERR_NOT_VORBIS;
  int rc;
  books.assign((size_t)p.read(8) + 1, Codebook());
  for (auto& b : books)
    if ((rc = b.init(p)) != NVH_OK) return rc;
  int times = (int)p.read(6) + 1;
  p.skip(16 * times);

  floors.assign((size_t)p.read(6) + 1, Floor());
  for (auto& f : floors) {
    int type = (int)p.read(16);  // Factory.cs:22-31
    if (type != 0 && type != 1) return NVH_ERR_INVALID_DATA;
    f.type = type;
    rc = type == 0 ? floor0_init(f.f0, p, block0, block1, books) : floor1_init(f.f1, p, (int)books.size());
    if (rc != NVH_OK) return rc;
  }
  residues.assign((size_t)p.read(6) + 1, Residue());
  for (auto& r : residues) {
    int type = (int)p.read(16);  // Factory.cs:48-58
    if (type < 0 || type > 2) return NVH_ERR_INVALID_DATA;
    if ((rc = residue_init(r, type, p, channels, books)) != NVH_OK) return rc;
  }
  mappings.assign((size_t)p.read(6) + 1, Mapping());
  for (auto& m : mappings) {
    if (p.read(16) != 0) return NVH_ERR_INVALID_DATA;  // Factory.cs:33-41
    if ((rc = mapping_init(m, p, channels, (int)floors.size(), (int)residues.size())) != NVH_OK) return rc;
  }
  windows.clear();
  modes.assign((size_t)p.read(6) + 1, Mode());
  for (auto& m : modes)
    if ((rc = mode_init(m, p, *this)) != NVH_OK) return rc;
  if (!p.read_bit()) return NVH_ERR_INVALID_DATA;  // StreamDecoder.cs:281
  mode_field_bits = ilog((int)modes.size() - 1);
  build_mdct_tables(block0, mdct[0]);
  build_mdct_tables(block1, mdct[1]);
  return NVH_OK;
}

}  // namespace nvh
