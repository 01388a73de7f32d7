// host_slab.cpp -- per-frame slabs written by the host packet parser's thread (see host_slab.h).
//
// Reference behaviour followed (file:line under /root/reference/NVorbis/):
//   Floor1.cs:196-216  Apply's walk over the posts in X order: a line from the last flagged post to the next, its far end
//                      clamped to n/2 (Math.Min(hx, n)), a flat run to n/2 behind the last one
//   Floor1.cs:224-297  UnwrapPosts (neighbour prediction, room arithmetic, step flags)
//   Floor1.cs:299-314  RenderPoint (int arithmetic, the product may wrap)
//   Floor1.cs:316-341  RenderLineMulti: y(x0 + k) = y0 + k b + sy floor(k r / adx) -- stored as a 32.32 fixed-point step
//   Residue0.cs:132-175, Residue2.cs:23-47  which bins a (stage, partition, channel) vector write touches
//   Mapping.cs:137-182  which coupling steps run (either channel executes), last step first
#include "host_slab.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace nvh {

namespace {

inline int render_point(int x0, int y0, int x1, int y1, int X) {  // Floor1.cs:299-314
  const int dy = y1 - y0, adx = x1 - x0;
  const int ady = dy < 0 ? -dy : dy;
  const int err = (int)((uint32_t)ady * (uint32_t)(X - x0));  // unchecked int multiply
  // truncating division, like C#'s -- through a double divide (exact here: |err| < 2^31, 0 < adx < 2^16, so a quotient that is not
  // an integer is at least 2^-16 away from one, far more than a double's rounding at this magnitude), ~3x cheaper than idiv
  const int off = (int)((double)err / (double)adx);
  return dy < 0 ? y0 - off : y0 + off;
}

// The digit bytes of one vector write (nvh_format.h: NVH_SLAB_RGEOM_DIGITS): n_ent entries of a book of dimension dim, appended to
// `out`; an entry that says "no vector" becomes dim bytes that point at the book's +0.0f slot.
inline void append_digits(const SlabSetup& X, uint32_t book, uint32_t dim, uint32_t lat_values, const uint16_t* e, uint32_t n_ent,
                          std::vector<uint8_t>& out) {
  const size_t o = out.size();
  out.resize(o + (size_t)n_ent * dim);
  uint8_t* d = out.data() + o;
  const uint8_t* tab = X.dig_tab.data() + X.dig_off[book];
  const uint8_t skip = (uint8_t)(lat_values * 4u);
  switch (dim) {
    case 2:
      for (uint32_t j = 0; j < n_ent; j++, d += 2)
        if (e[j] == NVH_ENTRY_SKIP) d[0] = d[1] = skip; else std::memcpy(d, tab + 2u * e[j], 2);
      break;
    case 4:
      for (uint32_t j = 0; j < n_ent; j++, d += 4)
        if (e[j] == NVH_ENTRY_SKIP) std::memset(d, skip, 4); else std::memcpy(d, tab + 4u * e[j], 4);
      break;
    case 8:
      for (uint32_t j = 0; j < n_ent; j++, d += 8)
        if (e[j] == NVH_ENTRY_SKIP) std::memset(d, skip, 8); else std::memcpy(d, tab + 8u * e[j], 8);
      break;
    default:
      for (uint32_t j = 0; j < n_ent; j++, d += dim)
        if (e[j] == NVH_ENTRY_SKIP) std::memset(d, skip, dim); else std::memcpy(d, tab + (size_t)dim * e[j], dim);
  }
}

// The entry section of a frame: uint16 entry numbers as the parser left them (padded with "no vector" to a whole 16-byte
// unit), or -- digit form -- the bytes the records' runs were collected into.
inline void put_entry_section(std::vector<SlabVec>& data, const FrameBatch& P, const NvhFrame& fr, bool dig, const std::vector<uint8_t>& dg) {
  const size_t e0 = data.size();
  if (dig) {
    data.resize(e0 + (dg.size() + 15) / 16);  // (zeroed by resize: padding bytes are never read)
    if (!dg.empty()) std::memcpy(&data[e0], dg.data(), dg.size());
    return;
  }
  const uint32_t ne = fr.ent_count, padded = (ne + 7) & ~7u;
  data.resize(e0 + padded / 8);
  uint16_t* dst = reinterpret_cast<uint16_t*>(&data[e0]);
  if (ne) std::memcpy(dst, P.entries.data() + fr.ent_begin, (size_t)ne * 2);
  for (uint32_t i = ne; i < padded; i++) dst[i] = (uint16_t)NVH_ENTRY_SKIP;
}

}  // namespace

int floor1_segments(const Floor1& f, const uint16_t* posts, int post_count, int half, SlabVec* seg, bool* fault) {
  const int cnt = (int)f.x_list.size();
  const int pc = post_count < cnt ? post_count : cnt;
  int final_y[NVH_MAX_POSTS];
  bool step[NVH_MAX_POSTS];
  for (int i = 0; i < NVH_MAX_POSTS; i++) { final_y[i] = 0; step[i] = false; }
  if (pc < 2) return 0;
  step[0] = step[1] = true;
  final_y[0] = posts[0];
  final_y[1] = posts[1];
  for (int i = 2; i < pc; i++) {
    const int lo = f.l_neigh[(size_t)i], hi = f.h_neigh[(size_t)i];
    const int predicted = render_point(f.x_list[(size_t)lo], final_y[lo], f.x_list[(size_t)hi], final_y[hi], f.x_list[(size_t)i]);
    const int val = posts[i];
    const int highroom = f.range - predicted, lowroom = predicted;
    const int room = (highroom < lowroom ? highroom : lowroom) * 2;
    if (val != 0) {
      step[lo] = step[hi] = step[i] = true;
      if (val >= room) final_y[i] = highroom > lowroom ? val - lowroom + predicted : predicted - val + highroom - 1;
      else final_y[i] = (val % 2) == 1 ? predicted - ((val + 1) / 2) : predicted + (val / 2);
    } else {
      final_y[i] = predicted;
    }
  }
  // the flagged posts in X order; the walk ends with the first line whose far end reaches n/2
  int px[NVH_MAX_POSTS + 1], py[NVH_MAX_POSTS + 1];
  int np = 0;
  px[np] = 0;  // lx = 0 whatever _xList[_sortIdx[0]] is (it is 0 for every valid header)
  py[np++] = final_y[0] * f.multiplier;
  bool reached = false;
  for (int i = 1; i < pc && !reached; i++) {
    const int idx = f.sort_idx[(size_t)i];
    if (idx >= pc || !step[idx]) continue;
    px[np] = f.x_list[(size_t)idx];
    py[np++] = final_y[idx] * f.multiplier;
    if (px[np - 1] >= half) reached = true;
  }
  if (!reached) {  // RenderLineMulti(lx, ly, n, ly)
    px[np] = half;
    py[np] = py[np - 1];
    np++;
  }
  const int ns = np - 1;
  for (int k = 0; k < ns; k++) {
    const int x0 = px[k], y0 = py[k], x1n = px[k + 1], y1 = py[k + 1];
    const int x1 = x1n < half ? x1n : half;
    const int dy = y1 - y0, adx = x1 - x0;
    if (adx <= 0 || x0 > 0xFFFF || x1n > 0xFFFF) return -1;  // not a curve the kernels can walk (headers with such X lists are refused)
    const uint32_t ady = (uint32_t)(dy < 0 ? -dy : dy);
    const uint32_t ab = ady / (uint32_t)adx, r = ady - ab * (uint32_t)adx;
    // ceil(r 2^32 / adx), r < adx <= 2^16: two 32-bit divisions (16 quotient bits each) instead of a 64-bit one -- this runs
    // ~60 times per stereo frame on the parser's thread
    uint64_t frac;
    {
      const uint32_t udx = (uint32_t)adx;
      const uint32_t n1 = r << 16, q1 = n1 / udx, r1 = n1 - q1 * udx;   // r < 2^16
      const uint32_t n2 = r1 << 16, q2 = n2 / udx, r2 = n2 - q2 * udx;  // r1 < adx <= 2^16
      frac = (((uint64_t)q1 << 16) | (uint64_t)q2) + (r2 ? 1u : 0u);
    }
    uint64_t F = ((uint64_t)ab << 32) + frac;
    if (dy < 0) F = 0ull - F;
    seg[k].x = (uint32_t)x0 | ((uint32_t)x1n << 16);
    seg[k].y = (uint32_t)y0;
    seg[k].z = (uint32_t)F;
    seg[k].w = (uint32_t)(F >> 32);
    // inverse_dB_table[y] (Floor1.cs:330,339) throws outside 0..255; the line is monotone, so its ends decide
    const int tl = adx - 1;
    const int b = dy < 0 ? -(int)ab : (int)ab;
    const int yl = y0 + b * tl + (dy < 0 ? -1 : 1) * (int)((r * (uint32_t)tl) / (uint32_t)adx);  // r, tl < 2^16
    if (y0 < 0 || y0 > 255 || yl < 0 || yl > 255) *fault = true;
  }
  return ns;
}

// A book the slab walks can take: a lattice book (digits or entry numbers) or one with an explicit table (entry numbers, the
// component gathered from the VQ pool); a book without a lookup (map type 0) has no vectors to add.
static inline bool book_has_vectors(const NvhDevBook& bk) { return bk.lat_values != 0 || bk.tab_off != 0xFFFFFFFFu; }

bool residue_alias_b1(const Setup& S, const SlabSetup& X, const Residue& r) {
  if (r.type != 2 || r.real_channels < 3 || r.real_channels > NVH_SLAB_MAX_CH) return false;
  if (r.begin % r.real_channels == 0 && r.partition_size % r.real_channels == 0) return false;  // no aliasing at all
  if (r.partition_size < 2 * r.real_channels || r.partition_size > 4096) return false;
  uint64_t max_div = (uint64_t)std::max(r.partition_size, r.real_channels);
  for (int c = 0; c < r.classifications && c < NVH_MAX_CLASSES; c++)
    for (int k = 0; k < NVH_MAX_STAGES; k++) {
      const int b = r.books[c][k];
      if (b < 0) continue;
      const NvhDevBook& bk = X.books[(size_t)b];
      if (!book_has_vectors(bk) || bk.dim == 0 || (bk.dim & 1u) || (uint32_t)r.partition_size % bk.dim != 0) return false;
      max_div = std::max<uint64_t>(max_div, bk.dim);
    }
  // the walk's reciprocal multiplies are exact while index * divisor < 2^32 (nvh_setup.hip: NvhDevResidue::fast -- the same
  // bound, here so that a stream without a device context classifies its residues exactly as the device build does)
  const uint64_t max_index = (uint64_t)(S.block1 / 2 + r.partition_size) * (uint64_t)r.real_channels;
  return max_index * max_div < 0x100000000ull;
}

bool residue_pair_ok(const Setup& S, const SlabSetup& X, const Residue& r) {
  if (r.type == 0 || r.partition_size < 2 || (r.partition_size & 1) || r.partition_size > 4096 || r.real_channels < 1) return false;
  if (r.type == 2 && (r.begin % r.real_channels != 0 || r.partition_size % r.real_channels != 0)) return false;  // aliasing partitions
  uint64_t max_div = (uint64_t)std::max(r.partition_size, r.real_channels);
  for (int c = 0; c < r.classifications && c < NVH_MAX_CLASSES; c++)
    for (int k = 0; k < NVH_MAX_STAGES; k++) {
      const int b = r.books[c][k];
      if (b < 0) continue;
      const NvhDevBook& bk = X.books[(size_t)b];
      if (!book_has_vectors(bk) || bk.dim == 0 || (bk.dim & 1u) || (uint32_t)r.partition_size % bk.dim != 0) return false;
      max_div = std::max<uint64_t>(max_div, bk.dim);
    }
  // (the reciprocal multiplies of the walk are exact while index * divisor < 2^32: nvh_setup.hip, NvhDevResidue::fast)
  const uint64_t max_index = (uint64_t)(S.block1 / 2 + r.partition_size) * (uint64_t)r.real_channels;
  return max_index * max_div < 0x100000000ull;
}

void classify_residues(const Setup& S, SlabSetup& X, bool no_pair) {
  const size_t n = S.residues.size();
  X.residue_pair.assign(n, 0);
  X.residue_b1.assign(n, 0);
  X.residue_general.assign(n, 0);
  for (size_t i = 0; i < n; i++) {
    const Residue& r = S.residues[i];
    // (Residue2 over more than two channels: a lane of the pair walk owns two bins of every channel)
    const bool whole_groups = !(r.type == 2 && r.real_channels > 2) || r.partition_size % (2 * r.real_channels) == 0;
    X.residue_pair[i] = (!no_pair && whole_groups && residue_pair_ok(S, X, r)) ? 1 : 0;
    X.residue_b1[i] = (!no_pair && !X.residue_pair[i] && residue_alias_b1(S, X, r)) ? 1 : 0;
    X.residue_general[i] = (!no_pair && residue_general_ok(S, X, r)) ? 1 : 0;
  }
  // the digit form of the entry sections: every book of every residue the slab kernels can take has its digit table
  X.digits_ok = X.val_off.size() == S.books.size() && S.books.size() > 0;
  for (size_t i = 0; i < n && X.digits_ok; i++) {
    if (!X.residue_pair[i] && !X.residue_b1[i] && !X.residue_general[i]) continue;
    const Residue& r = S.residues[i];
    for (int c = 0; c < r.classifications && c < NVH_MAX_CLASSES; c++)
      for (int k = 0; k < NVH_MAX_STAGES; k++) {
        const int b = r.books[c][k];
        if (b >= 0 && ((size_t)b >= X.dig_off.size() || X.dig_off[(size_t)b] == 0xFFFFFFFFu)) X.digits_ok = false;
      }
  }
  if (X.pool_words > (size_t)NVH_SLAB_MAX_LAT_OFF) X.digits_ok = false;  // a record addresses the pools with 12 bits of words
  if (!X.digits_ok) {
    // No slab of this setup takes the digit form: the value pool is not part of the kernels' constants block (it would be
    // staged into LDS by every workgroup for nothing and count against the LDS limit of slab_size_ok).
    X.val_pool.clear();
    X.pool_words = X.lattice.size();
  }
}

uint32_t residue_max_span(const Setup& S, const Residue& r) {
  uint32_t span = (uint32_t)(r.partition_size > 0 ? r.partition_size : 0);
  if (r.type == 0 || r.partition_size <= 0) return span;
  for (int c = 0; c < r.classifications && c < NVH_MAX_CLASSES; c++)
    for (int k = 0; k < NVH_MAX_STAGES; k++) {
      const int b = r.books[c][k];
      if (b < 0 || (size_t)b >= S.books.size()) continue;
      const uint32_t dim = (uint32_t)S.books[(size_t)b].dimensions;
      if (dim > 0) span = std::max(span, ((uint32_t)r.partition_size + dim - 1) / dim * dim);
    }
  return span;
}

bool residue_general_ok(const Setup& S, const SlabSetup& X, const Residue& r) {
  (void)S;
  const int rch = r.type == 2 ? r.real_channels : 1;
  if (rch < 1 || r.real_channels > NVH_SLAB_MAX_CH || r.partition_size < 2 || r.partition_size > 4096) return false;
  if (r.type == 2 && rch > 1 && (r.begin % rch != 0 || r.partition_size % rch != 0) && r.partition_size < 2 * rch) return false;
  const uint32_t psz = (uint32_t)r.partition_size;
  const uint32_t psz_magic = (uint32_t)((0x100000000ull + psz - 1) / psz);
  for (int c = 0; c < r.classifications && c < NVH_MAX_CLASSES; c++)
    for (int k = 0; k < NVH_MAX_STAGES; k++) {
      const int b = r.books[c][k];
      if (b < 0) continue;
      const NvhDevBook& bk = X.books[(size_t)b];
      if (!book_has_vectors(bk) || bk.lat_values > 0xFFu || bk.dim == 0 || bk.dim > 16u) return false;
      // A dimension that does not divide the partition: a Residue0 adds partition_size / dim whole steps (Residue0.cs:180-201: the
      // walk's geometry needs dim | size there); Residue1 / Residue2 add whole entries, the last one running over into the next
      // partition's elements (vector overrun, round 6: the bin walk merges the two partitions that touch a bin anyway) -- by
      // less than a partition, so that at most two partitions touch a bin.
      const uint32_t span = (psz + bk.dim - 1) / bk.dim * bk.dim;
      if (r.type == 0 ? psz % bk.dim != 0 : span - psz >= psz) return false;
      // the walk's divisions: i / dim and ceil(size / dim) by the 16-bit reciprocal, and (Residue0) i dim / partition_size by the 32-bit one
      if (((psz * bk.dim_magic16) >> 16) != psz / bk.dim || (((psz + bk.dim - 1) * bk.dim_magic16) >> 16) != (psz + bk.dim - 1) / bk.dim) return false;
      for (uint32_t i = 0; i < span; i++) {
        if (((i * bk.dim_magic16) >> 16) != i / bk.dim) return false;
        if (r.type == 0 && (uint32_t)(((uint64_t)(i * bk.dim) * psz_magic) >> 32) != i * bk.dim / psz) return false;
      }
    }
  return true;
}

bool floor0_section_values(const Floor0& f, int slot, int half, float amp, const float* coeff, float* qk) {
  const std::vector<int>& bark = f.bark_map[slot];
  const std::vector<float>& wmap = f.w_map[slot];
  const int K = f.bark_map_size;
  for (int k = 0; k < K; k++) qk[k] = 0.0f;
  if ((int)bark.size() < half || (int)wmap.size() < half) return false;
  std::vector<float> c2((size_t)f.order > 0 ? (size_t)f.order : 1);
  for (int i = 0; i < f.order; i++) c2[(size_t)i] = 2.0f * (float)std::cos((double)coeff[i]);  // Floor0.cs:165-168
  std::vector<uint8_t> done((size_t)(K > 0 ? K : 1), 0);
  for (int i = 0; i < half; i++) {
    const int k = bark[(size_t)i];
    if (k < 0 || k >= half || k >= K) return false;  // wMap[k] out of range (the Bark map's last entry is never written: quirk)
    if (done[(size_t)k]) continue;
    done[(size_t)k] = 1;
    float p = .5f, q = .5f;
    const float w = wmap[(size_t)k];
    int j;
    for (j = 1; j < f.order; j += 2) {
      q = q * (w - c2[(size_t)j - 1]);
      p = p * (w - c2[(size_t)j]);
    }
    if (j == f.order) {  // odd order
      q = q * (w - c2[(size_t)j - 1]);
      p = p * (p * (4.0f - w * w));
      q = q * q;
    } else {
      p = p * (p * (2.0f - w));
      q = q * (q * (2.0f + w));
    }
    q = amp / (float)std::sqrt((double)(p + q)) - (float)f.amp_ofs;
    q = (float)std::exp((double)(q * 0.11512925f));
    qk[k] = q;
  }
  return true;
}

namespace {

// The residue sections of a frame in the general form (NvhSlabHdr::group == 1): chains of every pass (heads, records), the
// frame's entries, and the group list -- one group per (pass, channel) of a per-channel residue, one per pass of a Residue2 --
// that tells the bin walk which chain covers which partition.  Returns NVH_OK with H's section fields set.
int residue_general(const Setup& S, const SlabSetup& X, const FrameBatch& P, const NvhFrame& fr, int nch, size_t base,
                    std::vector<SlabVec>& data, NvhSlabHdr& H, std::vector<uint32_t>& head_ops, std::vector<uint8_t>& dg) {
  const bool dig = X.digits_ok;
  dg.clear();
  const int npass = (int)(fr.pass_end - fr.pass_begin);
  const uint32_t nops = fr.op_count;
  const NvhResOp* ops = P.ops.data() + fr.op_begin;
  const uint16_t* links = P.op_link.data() + fr.op_begin;
  uint32_t nheads = 0;
  for (uint32_t o = 0; o < nops; o++) nheads += (links[o] & 0x8000u) ? 0u : 1u;
  const uint32_t off_heads = (uint32_t)(data.size() - base);
  const uint32_t off_rec = off_heads + ((nheads + 3) >> 2);
  data.resize(base + off_rec + ((nops + 1) >> 1));
  uint32_t* heads = reinterpret_cast<uint32_t*>(&data[base + off_heads]);
  uint32_t* recs = reinterpret_cast<uint32_t*>(&data[base + off_rec]);
  struct Group { uint32_t rbegin, psz, nparts, cover, geom, psz_magic, pchain_off; };
  std::vector<Group> groups;
  std::vector<uint16_t> pchain;
  uint32_t hk = 0, nrec = 0;
  for (int pi = 0; pi < npass; pi++) {
    const NvhResPass& gp = P.passes[fr.pass_begin + (uint32_t)pi];
    if (gp.residue < 0 || (size_t)gp.residue >= S.residues.size() || (size_t)gp.residue >= X.residue_general.size()) return NVH_ERR_RUNTIME;
    if (!X.residue_general[(size_t)gp.residue]) return NVH_ERR_UNSUPPORTED;
    const Residue& R = S.residues[(size_t)gp.residue];
    const uint32_t psz = (uint32_t)R.partition_size, rbegin = (uint32_t)R.begin;
    const int rch = R.type == 2 ? R.real_channels : 1;
    const int bs = R.type == 2 ? fr.n * R.real_channels : fr.n;  // Residue2.cs:16-21
    const int end = R.end < bs / 2 ? R.end : bs / 2;                 // Residue0.cs:122-123
    const int nn = end - R.begin;
    const uint32_t nparts = nn > 0 ? (uint32_t)(nn / R.partition_size) : 0u;
    // this pass's ops: [lo, hi) frame-relative
    const uint32_t lo = gp.op_begin[0] - fr.op_begin, hi = gp.op_begin[NVH_MAX_STAGES] - fr.op_begin;
    if (lo > hi || hi > nops) return NVH_ERR_RUNTIME;
    auto stage_of = [&](uint32_t o) {
      const uint32_t abs_o = fr.op_begin + o;
      unsigned st = 0;
      while (st + 1 < NVH_MAX_STAGES && abs_o >= gp.op_begin[st + 1]) ++st;
      return st;
    };
    const int pass_ch = R.channels;  // chains exist per (partition, channel); Residue2 decodes one interleaved vector: 1
    if (pass_ch < 1 || pass_ch > nch) return NVH_ERR_RUNTIME;
    const size_t pc0 = pchain.size();
    pchain.resize(pc0 + (size_t)pass_ch * nparts, (uint16_t)0xFFFFu);
    head_ops.clear();
    for (uint32_t o = lo; o < hi; o++)
      if (!(links[o] & 0x8000u)) head_ops.push_back(o);
    for (uint32_t o : head_ops) {
      const uint32_t part = ops[o].partition, c = ops[o].channel;
      if (part >= nparts || c >= (uint32_t)pass_ch) return NVH_ERR_RUNTIME;
      const uint32_t offset0 = rbegin + part * psz;
      const uint32_t xbase = rch > 1 ? offset0 / (uint32_t)rch : offset0;
      if (xbase > 0xFFFFu || nrec > 0xFFFFu || hk > 0xFFFEu) return NVH_ERR_UNSUPPORTED;
      pchain[pc0 + (size_t)c * nparts + part] = (uint16_t)hk;
      heads[hk++] = nrec | (xbase << 16);
      uint32_t q = o;
      for (;;) {
        if (q >= hi || nrec >= nops) return NVH_ERR_RUNTIME;
        const NvhResOp& op = ops[q];
        const NvhDevBook& bk = X.books[op.book];
        uint32_t rel = op.ent_off - fr.ent_begin, pool_off = bk.lat_off;
        if (dig) {  // the record's run of digit bytes: its entries' components (Residue0: size / dim entries; else ceil(size / dim))
          if (bk.dim == 0) return NVH_ERR_RUNTIME;
          const uint32_t n_ent = R.type == 0 ? psz / bk.dim : (psz + bk.dim - 1) / bk.dim;
          if ((R.type == 0 && psz % bk.dim != 0) || rel + n_ent > fr.ent_count || (dg.size() & 1u)) return NVH_ERR_RUNTIME;
          rel = (uint32_t)(dg.size() >> 1);
          pool_off = X.val_off[op.book];
          append_digits(X, op.book, bk.dim, bk.lat_values, P.entries.data() + op.ent_off, n_ent, dg);
          if (dg.size() & 1u) dg.push_back(0);  // (a partition of odd size: runs start on even bytes)
        }
        if (rel > 0xFFFFu || pool_off > NVH_SLAB_MAX_LAT_OFF || bk.lat_values > 0xFFu || bk.dim > 31u || op.channel > 7u ||
            op.partition != part || op.channel != c)
          return NVH_ERR_UNSUPPORTED;
        const uint32_t l = links[q] & 0x7FFFu;
        const uint32_t rw[2] = {NVH_SLAB_REC(rel, bk.dim_magic16, pool_off, bk.lat_values, bk.dim, op.channel, stage_of(q), l != NVH_LINK_NONE)};
        recs[2 * nrec] = rw[0];
        recs[2 * nrec + 1] = rw[1];
        ++nrec;
        if (l == NVH_LINK_NONE) break;
        q = l;
      }
    }
    for (int c = 0; c < pass_ch; c++) {
      Group g;
      g.rbegin = rbegin; g.psz = psz; g.nparts = nparts;
      g.cover = (residue_max_span(S, R) + (uint32_t)rch - 1) / (uint32_t)rch;  // bins a partition's longest vector write touches
      g.geom = (uint32_t)R.type | ((uint32_t)rch << 4) | ((uint32_t)pi << 8) | ((uint32_t)(R.type == 2 ? 0 : c) << 12);
      g.psz_magic = (uint32_t)((0x100000000ull + psz - 1) / psz);
      g.pchain_off = (uint32_t)(pc0 + (size_t)c * nparts);
      if (g.pchain_off > 0xFFFFFFu || pi > 15) return NVH_ERR_UNSUPPORTED;
      groups.push_back(g);
    }
  }
  for (uint32_t i = nheads; i < ((nheads + 3) & ~3u); i++) heads[i] = 0;
  if (nrec != nops || hk != nheads) return NVH_ERR_RUNTIME;  // every op belongs to exactly one chain
  if (nrec & 1u) recs[2 * nrec] = recs[2 * nrec + 1] = 0;
  H.off_heads = (uint16_t)off_heads;
  H.nheads = (uint16_t)nheads;
  H.nrec = (uint16_t)nrec;
  H.off_rec = (uint16_t)off_rec;
  H.rgeom = (uint8_t)(1 | (1 << 4) | (dig ? NVH_SLAB_RGEOM_DIGITS : 0u));
  H.group = 1;
  H.off_ent = (uint16_t)(data.size() - base);
  put_entry_section(data, P, fr, dig, dg);
  // the group list: count | per group two 16-byte units | uint16 pchain[]
  const size_t g0 = data.size();
  if (g0 - base > 0xFFFFu) return NVH_ERR_UNSUPPORTED;
  H.lpc = (uint16_t)(g0 - base);
  H.lpc_magic = 0;
  data.resize(g0 + 1 + 2 * groups.size() + (pchain.size() + 7) / 8);
  uint32_t* w = reinterpret_cast<uint32_t*>(&data[g0]);
  w[0] = (uint32_t)groups.size(); w[1] = w[2] = w[3] = 0;
  for (size_t i = 0; i < groups.size(); i++) {
    uint32_t* q = w + 4 + 8 * i;
    q[0] = groups[i].rbegin; q[1] = groups[i].psz; q[2] = groups[i].nparts; q[3] = groups[i].cover;
    q[4] = groups[i].geom; q[5] = groups[i].psz_magic; q[6] = groups[i].pchain_off; q[7] = 0;
  }
  uint16_t* pc = reinterpret_cast<uint16_t*>(w + 4 + 8 * groups.size());
  for (size_t i = 0; i < ((pchain.size() + 7) & ~(size_t)7); i++) pc[i] = i < pchain.size() ? pchain[i] : (uint16_t)0xFFFFu;
  return NVH_OK;
}

}  // namespace

int build_slabs(const Setup& S, const SlabSetup& X, const FrameBatch& P, SlabBatch& out) {
  out.clear();
  const int nch = S.channels;
  if (nch > NVH_SLAB_MAX_CH) return NVH_ERR_UNSUPPORTED;
  const size_t nf = P.frames.size();
  out.first.reserve(nf + 1);
  out.data.reserve(nf * 256);
  std::vector<SlabVec> segs((size_t)NVH_MAX_POSTS + 2);
  std::vector<uint32_t> head_scratch;  // op index of every chain head of the frame, in the order the heads are stored
  std::vector<uint8_t> dig_scratch;    // digit form: the frame's digit bytes, in record order
  const bool dig = X.digits_ok;
  for (size_t f = 0; f < nf; f++) {
    const NvhFrame& fr = P.frames[f];
    const size_t base = out.data.size();
    out.first.push_back((uint32_t)base);
    NvhSlabHdr H;
    std::memset(&H, 0, sizeof H);
    H.off_heads = H.off_rec = H.off_ent = NVH_SLAB_HDR_VECS;
    H.vecs = NVH_SLAB_HDR_VECS;
    H.group = 2;
    H.frame = (uint32_t)f;
    for (int c = 0; c < NVH_SLAB_MAX_CH; c++) H.chan[c] = (uint32_t)NVH_SLAB_HDR_VECS << 16;
    out.data.resize(base + NVH_SLAB_HDR_VECS);
    auto put_header = [&]() {
      static_assert(sizeof(NvhSlabHdr) == NVH_SLAB_HDR_VECS * 16, "header size");
      std::memcpy(&out.data[base], &H, sizeof H);
      const uint32_t v = (uint32_t)(out.data.size() - base);
      if (v > out.max_vecs) out.max_vecs = v;
    };
    if (fr.n == 0) {
      put_header();
      continue;
    }
    const int half = fr.n >> 1;
    if (fr.mapping < 0 || (size_t)fr.mapping >= S.mappings.size()) return NVH_ERR_RUNTIME;
    const Mapping& mp = S.mappings[(size_t)fr.mapping];
    const NvhChan* chans = &P.chans[f * (size_t)nch];
    H.n = (uint16_t)fr.n;
    H.exec_mask = (uint8_t)(fr.exec_mask & 0xFFu);
    if (fr.mdct_slot) H.flags |= NVH_SLAB_MDCT_SLOT;
    bool fault = false, any_floor0 = false;
    // ---- floors ----
    for (int c = 0; c < nch; c++) {
      const NvhChan& cn = chans[c];
      const Floor& fl = S.floors[cn.floor];
      const uint32_t off = (uint32_t)(out.data.size() - base);
      if (fl.type == 0) {
        // Floor0: one value per Bark section, evaluated here; the kernel multiplies bin i by value[barkMap[i]]
        const int mode0 = cn.exec ? (cn.amp > 0.0f ? 3 : 2) : 0;
        int nv = 0;
        if (mode0 == 3) {
          const int K = fl.f0.bark_map_size, slot = fr.mdct_slot ? 1 : 0;
          nv = 1 + (K + 3) / 4;
          if (nv > 255 || cn.floor >= X.floor0_bark_off[slot].size()) return NVH_ERR_UNSUPPORTED;
          any_floor0 = true;
          const size_t s0 = out.data.size();
          out.data.resize(s0 + (size_t)nv);
          out.data[s0].x = X.floor0_bark_off[slot][cn.floor];
          out.data[s0].y = (uint32_t)K;
          out.data[s0].z = out.data[s0].w = 0;
          float* qk = reinterpret_cast<float*>(&out.data[s0 + 1]);
          for (int k = K; k < 4 * (nv - 1); k++) qk[k] = 0.0f;
          if (!floor0_section_values(fl.f0, slot, half, cn.amp, &P.coeffs[cn.data_off], qk)) fault = true;
        }
        H.chan[c] = (uint32_t)mode0 | ((uint32_t)nv << 8) | (off << 16);
        continue;
      }
      const int mode = cn.exec ? (cn.post_count > 0 ? 1 : 2) : 0;
      int ns = 0;
      if (mode == 1) {
        ns = floor1_segments(fl.f1, &P.posts[cn.data_off], cn.post_count, half, segs.data(), &fault);
        if (ns <= 0 || ns > 255) return NVH_ERR_UNSUPPORTED;
        out.data.insert(out.data.end(), segs.begin(), segs.begin() + ns);
        // segment index of every group of four bins: the last segment that starts at or before the group's first bin
        const int ngroups = half >> 2, padded = (ngroups + 15) & ~15;
        const size_t t0 = out.data.size();
        out.data.resize(t0 + (size_t)padded / 16);
        uint8_t* tab = reinterpret_cast<uint8_t*>(&out.data[t0]);  // (zeroed by resize, padding included)
        // groups [ceil(x_k / 4), ceil(x_{k+1} / 4)) belong to segment k: one run fill per segment
        int g_lo = 0;
        for (int k = 0; k < ns; k++) {
          int g_hi = k + 1 < ns ? (int)(((segs[(size_t)k + 1].x & 0xFFFFu) + 3u) >> 2) : ngroups;
          if (g_hi > ngroups) g_hi = ngroups;
          if (g_hi > g_lo) {
            std::memset(tab + g_lo, k, (size_t)(g_hi - g_lo));
            g_lo = g_hi;
          }
        }
      }
      H.chan[c] = (uint32_t)mode | ((uint32_t)ns << 8) | (off << 16);
    }
    if (fault) H.flags |= NVH_SLAB_FLOOR_FAULT;
    // ---- residue: chains of vector writes, chain-major ----
    const int npass = (int)(fr.pass_end - fr.pass_begin);
    int rtype = 0, rch = 1;
    unsigned rbegin_al = 0;
    H.off_heads = (uint16_t)(out.data.size() - base);
    bool general = npass > 1;
    dig_scratch.clear();
    if (npass == 1) {
      const size_t ri = (size_t)P.passes[fr.pass_begin].residue;
      general = ri < X.residue_general.size() && X.residue_general[ri] != 0 && !(ri < X.residue_b1.size() && X.residue_b1[ri]) &&
                !(ri < X.residue_pair.size() && X.residue_pair[ri]);
    }
    if (general) {
      const int rcg = residue_general(S, X, P, fr, nch, base, out.data, H, head_scratch, dig_scratch);
      if (rcg != NVH_OK) return rcg;
    } else if (npass == 1) {
      const NvhResPass& gp = P.passes[fr.pass_begin];
      const Residue& R = S.residues[(size_t)gp.residue];
      rtype = R.type;
      rch = R.real_channels;
      const unsigned psz = (unsigned)R.partition_size, rbegin = (unsigned)R.begin;
      rbegin_al = rbegin;
      // quirk B-1 (partitions sharing a bin): the kernel walks bin by bin (group 0), chains in partition order
      const bool bins = (size_t)gp.residue < X.residue_b1.size() && X.residue_b1[(size_t)gp.residue] != 0;
      unsigned group = (psz & 7u) == 0 ? 8u : 2u;
      if (rtype == 2 && rch > 2) group = (psz % (2u * (unsigned)rch)) == 0 ? 2u * (unsigned)rch : 0u;
      if (bins) group = 0;
      else if (group == 0 || (psz % group) != 0) return NVH_ERR_UNSUPPORTED;
      const unsigned lpc = bins ? 0u : psz / group;
      H.group = (uint8_t)group;
      H.lpc = (uint16_t)lpc;
      H.lpc_magic = lpc > 1 ? (uint32_t)((0x100000000ull + lpc - 1) / lpc) : 0u;
      const uint32_t nops = fr.op_count;
      const NvhResOp* ops = P.ops.data() + fr.op_begin;
      const uint16_t* links = P.op_link.data() + fr.op_begin;
      // heads in op order, each chain's records consecutive
      uint32_t nheads = 0, nrec = 0;
      for (uint32_t o = 0; o < nops; o++) nheads += (links[o] & 0x8000u) ? 0u : 1u;
      const uint32_t off_heads = (uint32_t)(out.data.size() - base);
      const uint32_t off_rec = off_heads + ((nheads + 3) >> 2);
      out.data.resize(base + off_rec + ((nops + 1) >> 1));
      uint32_t* heads = reinterpret_cast<uint32_t*>(&out.data[base + off_heads]);
      uint32_t* recs = reinterpret_cast<uint32_t*>(&out.data[base + off_rec]);  // two words per record
      // the cascade stage of op o: ops are stage-major, pass.op_begin[] (batch-wide op indices) delimits the stages
      auto stage_of = [&](uint32_t o) {
        const uint32_t abs_o = fr.op_begin + o;
        unsigned st = 0;
        while (st + 1 < NVH_MAX_STAGES && abs_o >= gp.op_begin[st + 1]) ++st;
        return st;
      };
      uint32_t hk = 0;
      std::vector<uint32_t>& head_ops = head_scratch;
      head_ops.clear();
      for (uint32_t o = 0; o < nops; o++)
        if (!(links[o] & 0x8000u)) head_ops.push_back(o);
      if (bins)  // (Residue2: one chain per partition)
        std::stable_sort(head_ops.begin(), head_ops.end(), [&](uint32_t a, uint32_t b) { return ops[a].partition < ops[b].partition; });
      for (uint32_t o : head_ops) {
        const unsigned offset0 = rbegin + (unsigned)ops[o].partition * psz;
        const unsigned xbase = (rtype == 2 && rch > 1) ? offset0 / (unsigned)rch : offset0;
        if (xbase > 0xFFFFu || nrec > 0xFFFFu) return NVH_ERR_UNSUPPORTED;
        heads[hk++] = nrec | (xbase << 16);
        uint32_t q = o;
        for (;;) {
          if (q >= nops || nrec >= nops) return NVH_ERR_RUNTIME;
          const NvhResOp& op = ops[q];
          const NvhDevBook& bk = X.books[op.book];
          uint32_t rel = op.ent_off - fr.ent_begin, pool_off = bk.lat_off;
          if (dig) {  // the record's run of digit bytes: partition_size / dim entries (residue_pair_ok / residue_alias_b1: dim | size)
            if (bk.dim == 0 || psz % bk.dim != 0 || rel + psz / bk.dim > fr.ent_count || (dig_scratch.size() & 1u)) return NVH_ERR_RUNTIME;
            rel = (uint32_t)(dig_scratch.size() >> 1);
            pool_off = X.val_off[op.book];
            append_digits(X, op.book, bk.dim, bk.lat_values, P.entries.data() + op.ent_off, psz / bk.dim, dig_scratch);
          }
          if (rel > 0xFFFFu || pool_off > NVH_SLAB_MAX_LAT_OFF || bk.lat_values > 0xFFu || bk.dim > 31u || op.channel > 7u ||
              op.partition != ops[o].partition)
            return NVH_ERR_UNSUPPORTED;
          const uint32_t l = links[q] & 0x7FFFu;
          const uint32_t rw[2] = {NVH_SLAB_REC(rel, bk.dim_magic16, pool_off, bk.lat_values, bk.dim, op.channel, stage_of(q), l != NVH_LINK_NONE)};
          recs[2 * nrec] = rw[0];
          recs[2 * nrec + 1] = rw[1];
          ++nrec;
          if (l == NVH_LINK_NONE) break;
          q = l;
        }
      }
      for (uint32_t i = nheads; i < ((nheads + 3) & ~3u); i++) heads[i] = 0;
      if (nrec != nops) return NVH_ERR_RUNTIME;  // every op belongs to exactly one chain
      if (nrec & 1u) recs[2 * nrec] = recs[2 * nrec + 1] = 0;
      H.nheads = (uint16_t)nheads;
      H.nrec = (uint16_t)nrec;
      H.off_rec = (uint16_t)off_rec;
    } else {
      H.off_rec = (uint16_t)(out.data.size() - base);
    }
    if (!general) H.rgeom = (uint8_t)(rtype | (rch << 4) | (dig ? NVH_SLAB_RGEOM_DIGITS : 0u));
    // ---- the frame's vector entries: entry numbers, or the digit bytes of the records' runs ----
    if (!general) {
      H.off_ent = (uint16_t)(out.data.size() - base);
      put_entry_section(out.data, P, fr, dig, dig_scratch);
    }
    // ---- bin-by-bin walk (quirk B-1): the residue's geometry and which chain belongs to which partition ----
    if (!general && npass == 1 && H.group == 0) {
      const NvhResPass& gp = P.passes[fr.pass_begin];
      const Residue& R = S.residues[(size_t)gp.residue];
      const int bs = fr.n * R.real_channels;                       // Residue2.cs:16-21
      const int end = R.end < bs / 2 ? R.end : bs / 2;             // Residue0.cs:122-123
      const int nn = end - R.begin;
      const uint32_t nparts = nn > 0 ? (uint32_t)(nn / R.partition_size) : 0u;
      const uint32_t psz = (uint32_t)R.partition_size, rchu = (uint32_t)R.real_channels;
      const uint32_t off_bins = (uint32_t)(out.data.size() - base);
      const size_t b0 = out.data.size();
      out.data.resize(b0 + 1 + (nparts + 7) / 8);
      uint32_t* prm = reinterpret_cast<uint32_t*>(&out.data[b0]);
      prm[0] = (uint32_t)R.begin; prm[1] = psz; prm[2] = nparts; prm[3] = (psz + rchu - 1) / rchu;  // bins a partition touches
      uint16_t* pchain = reinterpret_cast<uint16_t*>(&out.data[b0 + 1]);
      for (uint32_t i = 0; i < ((nparts + 7) & ~7u); i++) pchain[i] = 0xFFFFu;
      const uint32_t* heads = reinterpret_cast<const uint32_t*>(&out.data[base + H.off_heads]);
      const NvhResOp* ops = P.ops.data() + fr.op_begin;
      for (uint32_t k = 0; k < H.nheads; k++) {
        const uint32_t p = ops[head_scratch[k]].partition;
        if (p >= nparts) return NVH_ERR_RUNTIME;
        pchain[p] = (uint16_t)k;
      }
      (void)heads;
      H.lpc = (uint16_t)off_bins;
      H.lpc_magic = (uint32_t)((0x100000000ull + psz - 1) / psz);
    }
    const size_t vecs = out.data.size() - base;
    if (vecs > 0xFFFFu) return NVH_ERR_UNSUPPORTED;
    H.vecs = (uint16_t)vecs;
    // ---- inverse coupling: in the chain walk when one lane holds both channels of a bin, else passes, last step first ----
    const int csteps = (int)mp.coupling_angle.size();
    if (nch == 2 && csteps == 1 && npass == 1 && rtype == 2 && rch == 2 && !general) {
      if ((fr.exec_mask & 3u) != 0) {
        if (mp.coupling_magnitude[0] == 1) H.flags |= NVH_SLAB_MG1;
        H.flags |= NVH_SLAB_SWEEP_COUPLES;
      }
    } else if (csteps > 0) {
      if (csteps > NVH_SLAB_MAX_COUPLE) return NVH_ERR_UNSUPPORTED;
      unsigned word = 0, cnt = 0;
      for (int st = csteps - 1; st >= 0; --st) {
        const unsigned mg = (unsigned)mp.coupling_magnitude[(size_t)st], an = (unsigned)mp.coupling_angle[(size_t)st];
        if (((fr.exec_mask >> mg) | (fr.exec_mask >> an)) & 1u) {
          word |= (mg | (an << 3)) << (4 + 6 * cnt);
          ++cnt;
        }
      }
      if (cnt) {
        H.coupling = word | cnt;
        H.flags |= NVH_SLAB_COUPLE_PASS;
      }
    }
    if (nch <= 2 && !(H.flags & NVH_SLAB_COUPLE_PASS) && !any_floor0 &&
        (npass == 0 || (!general && H.group == 8 && (rbegin_al & ((rtype == 2 && rch == 2) ? 7u : 3u)) == 0)))
      H.flags |= NVH_SLAB_FUSE_FLOOR;
    // ---- paired emission (nvh_format.h: NVH_EMIT_*): what k_synth needs to know about the overlaps ----
    if (nch <= 2 && (fr.emit_flags & NVH_EMIT_CARRY_OUT)) {
      H.exec_mask |= NVH_SLABX_CARRY_OUT;
      H.chan[2] = fr.window_off;
    }
    if (nch <= 2 && (fr.emit_flags & NVH_EMIT_SELF_CARRY)) H.exec_mask |= NVH_SLABX_SELF_CARRY;
    if (nch <= 2 && (fr.emit_flags & NVH_EMIT_DONE)) H.exec_mask |= NVH_SLABX_DONE;
    if (nch <= 2 && (fr.emit_flags & (NVH_EMIT_SELF | NVH_EMIT_NEXT | NVH_EMIT_DONE))) {
      H.chan[2] = fr.window_off; H.chan[3] = fr.ov_window_off; H.chan[6] = (uint32_t)fr.out_pos;
      const bool nxt = (fr.emit_flags & NVH_EMIT_NEXT) && f + 1 < nf;
      H.chan[5] = NVH_SLAB_GEO(fr.ov_n, nxt ? P.frames[f + 1].n : 0, fr.start, fr.valid);
      if (fr.emit_flags & NVH_EMIT_SELF) H.flags |= NVH_SLAB_EMIT_SELF;
      if (nxt) {
        const NvhFrame& nx = P.frames[f + 1];
        H.chan[4] = nx.window_off; H.chan[7] = (uint32_t)nx.out_pos;
        H.flags |= NVH_SLAB_EMIT_NEXT;
      }
    }
    put_header();
  }
  out.first.push_back((uint32_t)out.data.size());
  return NVH_OK;
}


// The codebook directory as the device holds it (NvhDevBook), the lattice pool (per lattice book: its distinct component values,
// then the reciprocal magics of lat_values^i) and the pool of VQ lookup tables (Codebook.cs:222-283 built them: host_setup.cpp).
void build_book_directory(const Setup& S, SlabSetup& X, std::vector<float>& vq, std::vector<uint32_t>& lattice) {
  vq.clear();
  lattice.clear();
  std::vector<NvhDevBook>& books = X.books;
  books.assign(S.books.size(), NvhDevBook{});
  for (size_t i = 0; i < S.books.size(); i++) {
    const Codebook& b = S.books[i];
    books[i].lat_values = 0;
    books[i].lat_magic = 0;
    books[i].lat_off = 0;
    books[i].dim_magic16 = b.dimensions >= 1 ? (uint32_t)((65536u + (uint32_t)b.dimensions - 1u) / (uint32_t)b.dimensions) : 0u;
    // lattice fast path: digits via exact reciprocal multiplies (entry < 2^16, powers <= entries)
    if (b.lattice_values >= 1 && b.dimensions >= 1 && b.dimensions <= 16 && b.entries <= 0xFFFF) {
      bool ok = true;
      std::vector<uint32_t> magics;
      uint64_t pw = 1;
      for (int d = 0; d < b.dimensions && ok; d++) {
        if (pw > 0xFFFF) { ok = false; break; }
        magics.push_back(pw > 1 ? (uint32_t)((0x100000000ull + pw - 1) / pw) : 0u);  // 0: divisor 1
        pw *= (uint64_t)b.lattice_values;
      }
      // self-check against the table the reference algorithm builds
      for (int e = 0; ok && e < b.entries; e++) {
        int q = e;
        for (int d = 0; d < b.dimensions; d++) {
          uint32_t bits_t, bits_l;
          float tv = b.lookup[(size_t)e * b.dimensions + d], lv = b.lattice[(size_t)(q % b.lattice_values)];
          std::memcpy(&bits_t, &tv, 4);
          std::memcpy(&bits_l, &lv, 4);
          if (bits_t != bits_l) { ok = false; break; }
          q /= b.lattice_values;
        }
      }
      if (ok) {
        books[i].lat_values = (uint32_t)b.lattice_values;
        books[i].lat_magic = b.lattice_values > 1 ? (uint32_t)((0x100000000ull + (uint64_t)b.lattice_values - 1) / (uint64_t)b.lattice_values) : 0u;
        books[i].lat_off = (uint32_t)lattice.size();
        for (float v : b.lattice) {
          uint32_t bits;
          std::memcpy(&bits, &v, 4);
          lattice.push_back(bits);
        }
        lattice.insert(lattice.end(), magics.begin(), magics.end());
      }
    }
    books[i].entries = (uint32_t)b.entries;
    books[i].dim = (uint32_t)b.dimensions;
    books[i].dim_magic = b.dimensions > 1 ? (uint32_t)((0x100000000ull + (uint64_t)b.dimensions - 1) / (uint64_t)b.dimensions) : 0u;
    if (b.map_type == 0) {
      books[i].tab_off = 0xFFFFFFFFu;
    } else {
      books[i].tab_off = (uint32_t)vq.size();
      vq.insert(vq.end(), b.lookup.begin(), b.lookup.end());
      if (books[i].lat_values == 0) {
        // A book with an explicit table (Codebook.cs:262-281: lookup type 2, or type 1 with sequence_p): its place in the lattice
        // pool is ONE word, the table's offset in the VQ pool -- a slab record then says lat_values = 0 and points at that word,
        // and the walks of the synthesis kernels gather the component out of the table (kernels_synth.hip: table_value).
        books[i].lat_off = (uint32_t)lattice.size();
        lattice.push_back(books[i].tab_off);
      }
    }
  }
  // the digit form (host_slab.h): value pool behind the lattice pool, digit bytes of every entry
  X.val_pool.clear();
  X.dig_tab.clear();
  X.val_off.assign(S.books.size(), 0xFFFFFFFFu);
  X.dig_off.assign(S.books.size(), 0xFFFFFFFFu);
  const size_t val_base = lattice.empty() ? 1 : lattice.size();  // (the device image pads an empty pool to one word)
  for (size_t i = 0; i < S.books.size(); i++) {
    const Codebook& b = S.books[i];
    const uint32_t lv = books[i].lat_values;
    if (lv == 0 || lv > NVH_SLAB_MAX_DIGIT || b.dimensions < 1 || b.dimensions > 16) continue;
    X.val_off[i] = (uint32_t)(val_base + X.val_pool.size());
    for (float v : b.lattice) {
      uint32_t bits;
      std::memcpy(&bits, &v, 4);
      X.val_pool.push_back(bits);
    }
    X.val_pool.push_back(0u);  // +0.0f: where the bytes of a vector that was never added point
    X.dig_off[i] = (uint32_t)X.dig_tab.size();
    X.dig_tab.resize(X.dig_tab.size() + (size_t)b.entries * (size_t)b.dimensions);
    uint8_t* t = X.dig_tab.data() + X.dig_off[i];
    for (int e = 0; e < b.entries; e++) {
      uint32_t q = (uint32_t)e;
      for (int d = 0; d < b.dimensions; d++) {
        *t++ = (uint8_t)((q % lv) * 4u);
        q /= lv;
      }
    }
  }
  X.pool_words = val_base + X.val_pool.size();
}

}  // namespace nvh
