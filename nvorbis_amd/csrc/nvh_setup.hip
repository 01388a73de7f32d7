// nvh_setup.hip -- device images of a stream's setup: the synthesis tables (codebook lookup tables, lattice pool,
// floor / residue / mapping records, windows, IMDCT twiddles) and the tables of the GPU packet parser.
#include "nvh_internal.h"

// ------------------------------------------------------------------------------------------------
// stream: setup upload
// ------------------------------------------------------------------------------------------------

namespace {

struct ArenaBuilder {
  std::vector<uint8_t> bytes;
  size_t add(const void* src, size_t n, size_t align = 16) {
    size_t off = (bytes.size() + align - 1) / align * align;
    bytes.resize(off + n);
    if (n) std::memcpy(bytes.data() + off, src, n);
    return off;
  }
};

}  // namespace

int upload_setup(nvh_stream* s) {
  const nvh::Setup& S = s->setup;
  ArenaBuilder ab;
  // documented limits of this build
  if (S.channels > 255) return NVH_ERR_UNSUPPORTED;
  if (S.books.size() > 256) return NVH_ERR_UNSUPPORTED;

  // codebook directory + lattice pool + VQ table pool (host_slab.cpp: the host slab writer uses the same directory)
  std::vector<float> vq;
  std::vector<uint32_t> lattice;
  nvh::build_book_directory(S, s->shared->slab, vq, lattice);
  nvh::classify_residues(S, s->shared->slab, nvh_toggles().no_pair || lattice.size() > 0xFFFFu);
  s->shared->slab.lattice = lattice;
  s->shared->slab.vq = vq;
  const std::vector<NvhDevBook>& books = s->shared->slab.books;

  std::vector<int32_t> ipool;
  std::vector<float> fpool;
  bool slab_floor0_ok = true;
  for (int w = 0; w < 2; w++) s->shared->slab.floor0_bark_off[w].assign(S.floors.size(), 0xFFFFFFFFu);
  std::vector<NvhDevFloor> floors(S.floors.size());
  for (size_t i = 0; i < S.floors.size(); i++) {
    const nvh::Floor& f = S.floors[i];
    NvhDevFloor& d = floors[i];
    std::memset(&d, 0, sizeof d);
    d.type = f.type;
    if (f.type == 1) {
      int cnt = (int)f.f1.x_list.size();
      d.f1.x_count = cnt;
      d.f1.multiplier = f.f1.multiplier;
      d.f1.range = f.f1.range;
      int lim = cnt < NVH_MAX_POSTS ? cnt : NVH_MAX_POSTS;  // more than 64 posts faults at decode time (host parser)
      int levels = 0;
      for (int k = 0; k < lim; k++) {
        if (f.f1.x_list[k] > 0xFFFF) return NVH_ERR_UNSUPPORTED;
        d.f1.x_list[k] = (uint16_t)f.f1.x_list[k];
        d.f1.l_neigh[k] = (uint8_t)f.f1.l_neigh[k];
        d.f1.h_neigh[k] = (uint8_t)f.f1.h_neigh[k];
        d.f1.sort_idx[k] = (uint8_t)(f.f1.sort_idx[k] < NVH_MAX_POSTS ? f.f1.sort_idx[k] : 0);
        int lv = 0;
        if (k >= 2) {
          int a = d.f1.level[f.f1.l_neigh[k]], b = d.f1.level[f.f1.h_neigh[k]];
          lv = (a > b ? a : b) + 1;
        }
        d.f1.level[k] = (uint8_t)lv;
        if (lv + 1 > levels) levels = lv + 1;
      }
      for (int k = 0; k < lim; k++) {
        const int lo = d.f1.l_neigh[k], hi = d.f1.h_neigh[k];
        d.f1.x_lo[k] = d.f1.x_list[lo < lim ? lo : 0];
        d.f1.x_hi[k] = d.f1.x_list[hi < lim ? hi : 0];
        d.f1.x_sorted[k] = d.f1.x_list[d.f1.sort_idx[k]];
        const int adx = (int)d.f1.x_hi[k] - (int)d.f1.x_lo[k];
        d.f1.adx_magic[k] = (k >= 2 && adx > 0) ? 0xFFFFFFFFu / (uint32_t)adx : 0u;
      }
      d.f1.levels = levels;
    } else {
      d.f0.order = f.f0.order;
      d.f0.amp_ofs = f.f0.amp_ofs;
      d.f0.bark_map_size = f.f0.bark_map_size;
      if (f.f0.order > 255) return NVH_ERR_UNSUPPORTED;
      if (f.f0.bark_map_size > 1016) slab_floor0_ok = false;  // a slab's floor section holds the per-section values in <= 254 vectors
      for (int w = 0; w < 2; w++) {
        s->shared->slab.floor0_bark_off[w].resize(S.floors.size(), 0xFFFFFFFFu);
        s->shared->slab.floor0_bark_off[w][i] = (uint32_t)ipool.size();
        d.f0.bark_off[w] = (uint32_t)ipool.size();
        ipool.insert(ipool.end(), f.f0.bark_map[w].begin(), f.f0.bark_map[w].end());
        d.f0.wmap_off[w] = (uint32_t)fpool.size();
        fpool.insert(fpool.end(), f.f0.w_map[w].begin(), f.f0.w_map[w].end());
      }
    }
  }

  std::vector<NvhDevResidue> residues(S.residues.size());
  for (size_t i = 0; i < S.residues.size(); i++) {
    const nvh::Residue& r = S.residues[i];
    NvhDevResidue& d = residues[i];
    d.type = r.type;
    d.begin = r.begin;
    d.end = r.end;
    d.partition_size = r.partition_size;
    d.classifications = r.classifications;
    d.channels = r.channels;
    d.real_channels = r.real_channels;
    bool seq = false;
    if (r.type == 2 && (r.begin % r.real_channels != 0 || r.partition_size % r.real_channels != 0)) seq = true;
    for (int c = 0; c < r.classifications; c++)
      for (int k = 0; k < NVH_MAX_STAGES; k++) {
        int b = r.books[c][k];
        if (b < 0) continue;
        const nvh::Codebook& bk = S.books[(size_t)b];
        if (bk.entries > 0xFFFF) return NVH_ERR_UNSUPPORTED;  // entry stream is 16-bit (0xFFFF = skip)
        if (bk.dimensions > 0 && r.type != 0 && r.partition_size % bk.dimensions != 0) seq = true;  // vector overrun
      }
    d.sequential = seq ? 1 : 0;
    if (seq) s->shared->has_sequential = true;
    d.psize_magic = r.partition_size > 1 ? (uint32_t)((0x100000000ull + (uint64_t)r.partition_size - 1) / (uint64_t)r.partition_size) : 0u;
    d.rch_magic = r.real_channels > 1 ? (uint32_t)((0x100000000ull + (uint64_t)r.real_channels - 1) / (uint64_t)r.real_channels) : 0u;
    // reciprocal multiplies are exact while index * divisor < 2^32; indices stay below
    // (partitions per stage) * channels * partition_size <= block1/2 * channels (+ one partition of overrun)
    {
      uint64_t max_index = (uint64_t)(S.block1 / 2 + r.partition_size) * (uint64_t)(r.real_channels > 0 ? r.real_channels : 1);
      uint64_t max_div = (uint64_t)r.partition_size;
      if ((uint64_t)r.real_channels > max_div) max_div = (uint64_t)r.real_channels;
      for (int c = 0; c < r.classifications; c++)
        for (int k = 0; k < NVH_MAX_STAGES; k++)
          if (r.books[c][k] >= 0 && (uint64_t)S.books[(size_t)r.books[c][k]].dimensions > max_div)
            max_div = (uint64_t)S.books[(size_t)r.books[c][k]].dimensions;
      d.fast = (r.type != 0 && r.partition_size > 1 && max_index * max_div < 0x100000000ull) ? 1 : 0;
      // the pair path and the B-1 bin walk (host_slab.cpp: residue_pair_ok, residue_alias_b1; pair records pack LDS offsets /
      // bin indices into 16 bits and use a 16-bit reciprocal of the book dimension)
      // (the descriptor kernels' pair path peels lattice digits and nothing else: a residue with an explicit-table book, which the
      // slab kernels take since round 6, keeps their general path)
      bool all_lattice = true;
      for (int c = 0; c < r.classifications && c < NVH_MAX_CLASSES; c++)
        for (int k = 0; k < NVH_MAX_STAGES; k++)
          if (r.books[c][k] >= 0 && s->shared->slab.books[(size_t)r.books[c][k]].lat_values == 0) all_lattice = false;
      d.pair_path = (!nvh_toggles().no_pair && all_lattice && lattice.size() <= 0xFFFFu && nvh::residue_pair_ok(S, s->shared->slab, r)) ? 1 : 0;
      d.alias_b1 = (seq && d.fast != 0 && s->shared->slab.residue_b1[i]) ? 1 : 0;
      s->shared->slab.residue_b1[i] = (uint8_t)d.alias_b1;
      d.hp_magic = r.partition_size / 2 > 1 ? (uint32_t)((0x100000000ull + (uint64_t)(r.partition_size / 2) - 1) / (uint64_t)(r.partition_size / 2)) : 0u;
      d.pad[0] = d.pad[1] = 0;
    }
  }

  std::vector<uint8_t> coupling;
  std::vector<NvhDevMapping> mappings(S.mappings.size());
  for (size_t i = 0; i < S.mappings.size(); i++) {
    mappings[i].coupling_steps = (int32_t)S.mappings[i].coupling_angle.size();
    mappings[i].coupling_off = (uint32_t)coupling.size();
    for (size_t k = 0; k < S.mappings[i].coupling_angle.size(); k++) {
      coupling.push_back((uint8_t)S.mappings[i].coupling_magnitude[k]);
      coupling.push_back((uint8_t)S.mappings[i].coupling_angle[k]);
    }
  }
  if (coupling.empty()) coupling.push_back(0);
  if (vq.empty()) vq.push_back(0.0f);
  if (lattice.empty()) lattice.push_back(0u);
  if (ipool.empty()) ipool.push_back(0);
  if (fpool.empty()) fpool.push_back(0.0f);

  size_t o_vq = ab.add(vq.data(), vq.size() * sizeof(float));
  size_t o_lat = ab.add(lattice.data(), lattice.size() * sizeof(uint32_t));
  // constants block of the slab synthesis kernel (kernels_synth.hip): inverse_dB_table followed by the lattice pool, in whole
  // 16-byte units, fetched by LDS-DMA as one piece
  std::vector<uint32_t> synth_consts(256);
  {
    static const float db_table[256] = {
#include "floor1_db_table.inc"
    };
    std::memcpy(synth_consts.data(), db_table, sizeof db_table);
    synth_consts.insert(synth_consts.end(), lattice.begin(), lattice.end());
    // the value pool of the digit form (host_slab.h: SlabSetup::val_pool): a record's offset counts from the lattice pool's start
    const std::vector<uint32_t>& vp = s->shared->slab.val_pool;
    synth_consts.insert(synth_consts.end(), vp.begin(), vp.end());
    while (synth_consts.size() & 3u) synth_consts.push_back(0u);
  }
  size_t o_sc = ab.add(synth_consts.data(), synth_consts.size() * sizeof(uint32_t));
  size_t o_books = ab.add(books.data(), books.size() * sizeof(NvhDevBook));
  size_t o_floors = ab.add(floors.data(), floors.size() * sizeof(NvhDevFloor));
  size_t o_res = ab.add(residues.data(), residues.size() * sizeof(NvhDevResidue));
  size_t o_map = ab.add(mappings.data(), mappings.size() * sizeof(NvhDevMapping));
  size_t o_cpl = ab.add(coupling.data(), coupling.size());
  size_t o_win = ab.add(S.windows.data(), S.windows.size() * sizeof(float));
  // reciprocals of every possible floor segment length (kernels_spectrum.hip: floor_prepare)
  std::vector<uint32_t> recip((size_t)S.block1 / 2 + 1, 0u);
  for (size_t d = 1; d < recip.size(); d++) recip[d] = (uint32_t)(0xFFFFFFFFull / d);
  size_t o_recip = ab.add(recip.data(), recip.size() * sizeof(uint32_t));
  size_t o_ip = ab.add(ipool.data(), ipool.size() * sizeof(int32_t));
  size_t o_fp = ab.add(fpool.data(), fpool.size() * sizeof(float));
  size_t o_a[2], o_b[2], o_c[2], o_br[2], o_tw[2];
  for (int w = 0; w < 2; w++) {
    o_a[w] = ab.add(S.mdct[w].a.data(), S.mdct[w].a.size() * sizeof(float));
    o_b[w] = ab.add(S.mdct[w].b.data(), S.mdct[w].b.size() * sizeof(float));
    o_c[w] = ab.add(S.mdct[w].c.data(), S.mdct[w].c.size() * sizeof(float));
    o_br[w] = ab.add(S.mdct[w].bitrev.data(), S.mdct[w].bitrev.size() * sizeof(uint16_t));
    o_tw[w] = ab.add(S.mdct[w].tw.data(), S.mdct[w].tw.size() * sizeof(float));
  }

  s->has_floor0 = false;
  for (const auto& fl : S.floors) s->has_floor0 = s->has_floor0 || fl.type == 0;
  int rc = s->arena.reserve(ab.bytes.size());
  if (rc != NVH_OK) return rc;
  HIP_TRY(hipMemcpy(s->arena.p, ab.bytes.data(), ab.bytes.size(), hipMemcpyHostToDevice));
  const uint8_t* base = (const uint8_t*)s->arena.p;
  NvhDevSetup& D = s->dev;
  D.channels = S.channels;
  D.block0 = S.block0;
  D.block1 = S.block1;
  D.nbooks = (int32_t)S.books.size();
  D.vq = (const float*)(base + o_vq);
  D.lattice = (const uint32_t*)(base + o_lat);
  D.lattice_words = (int32_t)lattice.size();
  s->shared->synth_consts = (const uint4*)(base + o_sc);
  s->shared->synth_const_vecs = (int)(synth_consts.size() / 4);
  s->shared->max_posts = 0;
  for (const auto& fl : S.floors)
    if (fl.type == 1) s->shared->max_posts = std::max(s->shared->max_posts, (int)fl.f1.x_list.size());
  {
    bool ok = S.channels <= 2 && !s->has_floor0;
    for (const nvh::Mapping& m : S.mappings) ok = ok && m.coupling_angle.size() <= 1;
    D.fused_tail_ok = ok ? 1 : 0;
    bool all_pairs = true;
    for (const NvhDevResidue& r : residues) all_pairs = all_pairs && r.pair_path != 0;
    s->fast_spectrum = ok && all_pairs;
    // slab synthesis kernels (kernels_synth.hip; nvh_launch.hip: slab_path)
    // every residue on the pair path, aliasing in the B-1 way only, or inside the general bin walk's contract; a stream that
    // needs the general walk anywhere (or has several submaps: more than one residue pass per frame) runs the wide kernel
    bool slab_res = true, general = false;
    const nvh::SlabSetup& X = s->shared->slab;
    for (size_t i = 0; i < residues.size(); i++) {
      slab_res = slab_res && (X.residue_pair[i] != 0 || X.residue_b1[i] != 0 || X.residue_general[i] != 0);
      general = general || (X.residue_pair[i] == 0 && X.residue_b1[i] == 0);
    }
    for (const nvh::Mapping& m : S.mappings)
      if (m.submap_floor.size() > 1) {
        general = true;
        for (int ri : m.submap_residue) slab_res = slab_res && ri >= 0 && (size_t)ri < residues.size() && X.residue_general[(size_t)ri] != 0;
      }
    s->shared->slab_general = general;
    bool slab_ok = slab_floor0_ok && slab_res && S.channels <= NVH_SLAB_MAX_CH && S.block0 >= 256 && S.block1 <= 8192 &&
                   s->shared->synth_consts != nullptr;
    for (const nvh::Mapping& m : S.mappings) slab_ok = slab_ok && m.coupling_angle.size() <= (size_t)NVH_SLAB_MAX_COUPLE;
    slab_ok = slab_ok && lattice.size() <= (size_t)NVH_SLAB_MAX_LAT_OFF;  // a record addresses the lattice pool with 12 bits
    s->shared->slab_setup_ok = slab_ok;
  }
  D.books = (const NvhDevBook*)(base + o_books);
  D.floors = (const NvhDevFloor*)(base + o_floors);
  D.residues = (const NvhDevResidue*)(base + o_res);
  D.mappings = (const NvhDevMapping*)(base + o_map);
  D.coupling = base + o_cpl;
  D.windows = (const float*)(base + o_win);
  D.recip = (const uint32_t*)(base + o_recip);
  D.ipool = (const int32_t*)(base + o_ip);
  D.fpool = (const float*)(base + o_fp);
  for (int w = 0; w < 2; w++) {
    D.mdct_a[w] = (const float*)(base + o_a[w]);
    D.mdct_b[w] = (const float*)(base + o_b[w]);
    D.mdct_c[w] = (const float*)(base + o_c[w]);
    D.mdct_br[w] = (const uint16_t*)(base + o_br[w]);
    D.mdct_tw[w] = (const float*)(base + o_tw[w]);
  }
  s->shared->dev_copy.pool = s->arena.pool;
  if ((rc = s->shared->dev_copy.reserve(sizeof(NvhDevSetup))) != NVH_OK) return rc;
  HIP_TRY(hipMemcpy(s->shared->dev_copy.p, &D, sizeof D, hipMemcpyHostToDevice));
  return NVH_OK;
}

// ------------------------------------------------------------------------------------------------
// batches
// ------------------------------------------------------------------------------------------------

// Moves s->pending into `b` (device resident) and advances the stream's batch boundary.
// Device tables of the GPU packet parser (kernels_parse.hip) and the worst-case slab capacities of this setup.
// Streams outside its limits (Floor0, > 8 channels, ...) simply keep the host parser.
int upload_parse_tables(nvh_stream* s) {
  const nvh::Setup& S = s->setup;
  SharedSetup& sh = *s->shared;
  sh.gpu_parse_ok = false;
  if (S.channels > NVH_PARSE_MAX_CH || S.books.size() > 256) return NVH_OK;
  for (const nvh::Floor& f : S.floors)
    if (f.type != 1) return NVH_OK;
  for (const nvh::Mapping& m : S.mappings)
    if (m.submap_floor.size() > NVH_PARSE_MAX_SUBMAPS || m.coupling_angle.size() > NVH_PARSE_MAX_COUPLING) return NVH_OK;

  std::vector<NvhPBook> books(S.books.size());
  std::vector<uint32_t> prefix;
  std::vector<NvhPOverflow> overflow;
  for (size_t i = 0; i < S.books.size(); i++) {
    const nvh::Codebook& b = S.books[i];
    NvhPBook& d = books[i];
    std::memset(&d, 0, sizeof d);
    if (b.entries > 0xFFFFFF || b.prefix_bits > 16 || b.max_bits > 32 || b.dimensions > 0xFFFF) return NVH_OK;
    d.prefix_off = (uint32_t)prefix.size();
    d.ovf_off = (uint32_t)overflow.size();
    d.entries = (uint32_t)b.entries;
    d.dims = (uint16_t)b.dimensions;
    d.prefix_bits = (uint8_t)b.prefix_bits;
    d.max_bits = (uint8_t)b.max_bits;
    d.has_tree = b.has_tree ? 1 : 0;
    d.has_overflow = b.has_overflow ? 1 : 0;
    d.dim_magic = b.dimensions > 1 ? (uint32_t)((0x100000000ull + (uint64_t)b.dimensions - 1) / (uint64_t)b.dimensions) : 0u;
    if (i < sh.slab.books.size()) {
      const NvhDevBook& db = sh.slab.books[i];
      d.slab_lat = db.lat_off | (db.lat_values << 16);
      d.slab_dm16 = db.dim_magic16;
    }
    // prefix[slot]: a short code, or (for slots only longer codes start with) that slot's group of overflow nodes
    //   present: (value << 8) | 0x80 | length        absent: (group begin << 8) | group count (0x7F = scan the whole list)
    for (size_t k = 0; k < b.prefix.size(); k++) {
      const nvh::HuffNode& n = b.prefix[k];
      if (n.present) {
        if (n.length < 0 || n.length > 0x7F || n.value < 0 || n.value > 0xFFFFFF) return NVH_OK;
        prefix.push_back(((uint32_t)n.value << 8) | 0x80u | (uint32_t)n.length);
      } else {
        uint32_t g = b.has_overflow && k < b.slot_group.size() ? b.slot_group[k] : 0u;
        uint32_t cnt = g & 0xFFu, beg = g >> 8;
        if (cnt >= 0x7Fu || beg > 0xFFFFFFu) {  // oversized group (or the host's own fallback marker): plain scan
          cnt = 0x7Fu;
          beg = 0;
        }
        prefix.push_back((beg << 8) | cnt);
      }
    }
    if (b.prefix.empty()) prefix.push_back(0u);  // has_tree == false: never indexed, keeps offsets valid
    // overflow pool of this book: the whole list in the reference's order, then the same nodes grouped by slot
    auto put = [&](const nvh::HuffNode& n) {
      NvhPOverflow o;
      o.bits = (uint32_t)n.bits;
      o.mask = (uint32_t)n.mask;
      o.value = (uint32_t)n.value;
      o.length = (uint32_t)n.length;
      overflow.push_back(o);
    };
    for (const nvh::HuffNode& n : b.overflow) put(n);
    for (const nvh::HuffNode& n : b.overflow_grouped) put(n);
    d.ovf_count = (uint32_t)b.overflow.size();
  }
  // LDS image: the decode maps and visit descriptors, the residue VQ books, then the class and floor books, while they fit
  std::vector<uint32_t> lds_image, dm_lds, vis_lds, sub_image;
  {
    const size_t budget = 16 * 1024;  // words (64 KB; + <= 8 KB of book / floor / residue / mapping records).  The residue books first:
                                      // they fill 13 k words for a libvorbis setup, and one of them out of LDS costs more than
                                      // everything else in it saves (round 6, class and floor books in front with 17 k words: the
                                      // entry loops of a packet 200 k -> 980 k cycles, a parse 0.43 -> 0.74 ms); what is left goes to
                                      // the class and floor books, whose symbols come through L2 at ~1-2 k cycles each otherwise.
    for (auto& d : books) d.lds_off = 0xFFFFFFFFu;
    // In front of the books: the residues' class decode maps (Residue0.cs:60-76: class word -> the classes of its partitions), a
    // few hundred words.  Every (stage, partition, channel) visit of the walk looks its class up; from the int pool in global
    // memory that lookup made the visit wait for its memory round trip AND for every store still in flight (one counter serves
    // both): half of a packet's parse time in the wave-uniform form.
    dm_lds.assign(S.residues.size(), 0xFFFFFFFFu);
    vis_lds.assign(S.residues.size(), 0xFFFFFFFFu);
    for (size_t i = 0; i < S.residues.size(); i++) {
      const std::vector<int>& dm = S.residues[i].decode_map;
      if (dm.empty() || lds_image.size() + dm.size() > 1024) continue;
      dm_lds[i] = (uint32_t)lds_image.size();
      for (int v : dm) lds_image.push_back((uint32_t)v);
    }
    // ... and room for the visit descriptors (nvh_parse_format.h: NVH_PVIS_*), 16-byte aligned; filled in below, when the books
    // have their places
    for (size_t i = 0; i < S.residues.size(); i++) {
      const size_t nw = (size_t)std::min(S.residues[i].classifications, NVH_MAX_CLASSES) * NVH_MAX_STAGES * 4;
      while (lds_image.size() & 3u) lds_image.push_back(0u);
      if (nw == 0 || lds_image.size() + nw > 3 * 1024) continue;
      vis_lds[i] = (uint32_t)lds_image.size();
      lds_image.resize(lds_image.size() + nw, 0u);
    }
    std::vector<int> order;
    std::vector<char> seen(S.books.size(), 0);
    auto want = [&](int b) {
      if (b >= 0 && b < (int)S.books.size() && !seen[(size_t)b]) {
        seen[(size_t)b] = 1;
        order.push_back(b);
      }
    };
    for (const nvh::Residue& r : S.residues)
      for (int c = 0; c < r.classifications && c < NVH_MAX_CLASSES; c++)
        for (int k = 0; k < NVH_MAX_STAGES; k++) want(r.books[c][k]);
    for (const nvh::Residue& r : S.residues) want(r.class_book);
    for (const nvh::Floor& fl : S.floors)
      for (int k = 0; k < 16; k++) {
        want(fl.f1.class_masterbook[k]);
        for (int j = 0; j < 8; j++) want(fl.f1.subclass_book[k][j]);
      }
    for (int b : order) {
      const size_t n = S.books[(size_t)b].prefix.size();
      if (n == 0 || lds_image.size() + n > budget) continue;
      books[(size_t)b].lds_off = (uint32_t)lds_image.size();
      lds_image.insert(lds_image.end(), prefix.begin() + books[(size_t)b].prefix_off, prefix.begin() + books[(size_t)b].prefix_off + n);
    }
    // ... and behind them the grouped overflow nodes of those books, 8 bytes each: a code longer than the prefix costs a scan of
    // its slot's group, a chain of dependent loads -- from LDS instead of from L2 (the C5 writer's packets, whose entries are
    // drawn uniformly, take this path for 25-96 % of their symbols)
    for (auto& d : books) d.ovf_lds = 0xFFFFFFFFu;
    const size_t node_budget = lds_image.size() + 5 * 1024;  // words (20 KB)
    for (int b : order) {
      NvhPBook& d = books[(size_t)b];
      const nvh::Codebook& cb = S.books[(size_t)b];
      if (d.lds_off == 0xFFFFFFFFu || !cb.has_overflow || cb.overflow_grouped.empty()) continue;
      bool ok = lds_image.size() + 2 * cb.overflow_grouped.size() <= node_budget;
      for (const nvh::HuffNode& n : cb.overflow_grouped)
        ok = ok && n.length >= 1 && n.length <= 31 && n.value >= 0 && n.value <= 0xFFFFFF && (uint32_t)n.mask == (1u << n.length) - 1u;
      if (!ok) continue;
      d.ovf_lds = (uint32_t)lds_image.size();
      for (const nvh::HuffNode& n : cb.overflow_grouped) {
        lds_image.push_back((uint32_t)n.bits);
        lds_image.push_back(((uint32_t)n.value << 8) | (uint32_t)n.length);
      }
    }
    if (lds_image.empty()) lds_image.push_back(0u);
    // Second-level tables for those books (k_parse_slab_f only, its own LDS block behind the records): per book a directory with one
    // word per grouped node -- the word at a group's first node says where the group's table lies and how many bits index it -- and
    // per group a table over the bits behind the prefix, as wide as the group's longest code: entry = (value << 8) | 0x80 | the code's
    // whole length, 0 = no code.  A long code then is two LDS reads instead of a scan of its group (books of a libvorbis setup:
    // up to 27 nodes per group, 3.5 k table entries in all).  Filled in the list's order, an entry keeps its first code: the node
    // the reference's scan would stop at (Codebook.cs:307-318) for every bit pattern.
    for (auto& d : books) d.sub_dir = 0xFFFFFFFFu;
    for (int b : order) {
      NvhPBook& d = books[(size_t)b];
      const nvh::Codebook& cb = S.books[(size_t)b];
      if (d.ovf_lds == 0xFFFFFFFFu || cb.slot_group.size() != cb.prefix.size()) continue;
      const uint32_t pb = d.prefix_bits;
      const size_t dir_off = sub_image.size();
      sub_image.resize(dir_off + cb.overflow_grouped.size(), 0u);
      bool ok = pb >= 1 && pb <= 24;
      for (size_t k = 0; ok && k < cb.prefix.size(); k++) {
        if (cb.prefix[k].present) continue;
        const uint32_t g = cb.slot_group[k], cnt = g & 0xFFu, beg = g >> 8;
        if (cnt == 0) continue;
        if (cnt >= 0x7Fu || beg + cnt > cb.overflow_grouped.size()) { ok = false; break; }
        uint32_t gmax = 0;
        for (uint32_t j = 0; j < cnt; j++) gmax = std::max<uint32_t>(gmax, (uint32_t)cb.overflow_grouped[beg + j].length);
        if (gmax <= pb || gmax - pb > 12 || gmax > 32) { ok = false; break; }
        const uint32_t sbits = gmax - pb;
        const size_t sub_off = sub_image.size();
        if (sub_off + ((size_t)1 << sbits) > 6 * 1024 || sub_off > 0xFFFFFFu) { ok = false; break; }
        sub_image.resize(sub_off + ((size_t)1 << sbits), 0u);
        for (uint32_t j = 0; j < cnt; j++) {
          const nvh::HuffNode& n = cb.overflow_grouped[beg + j];
          const uint32_t len = (uint32_t)n.length, rl = len - pb;
          if (len <= pb || ((uint32_t)n.bits & ((1u << pb) - 1u)) != (uint32_t)k || n.value < 0 || n.value > 0xFFFFFF) { ok = false; break; }
          const uint32_t rest = (len >= 32 ? (uint32_t)n.bits : ((uint32_t)n.bits & ((1u << len) - 1u))) >> pb;
          for (uint32_t hi = 0; hi < (1u << (sbits - rl)); hi++) {
            uint32_t& e = sub_image[sub_off + ((hi << rl) | rest)];
            if (e == 0u) e = ((uint32_t)n.value << 8) | 0x80u | len;
          }
        }
        sub_image[dir_off + beg] = (uint32_t)sub_off | (sbits << 24);
      }
      if (!ok) {
        sub_image.resize(dir_off);
        continue;
      }
      d.sub_dir = (uint32_t)dir_off;
    }
    if (sub_image.empty()) sub_image.push_back(0u);
    // the visit descriptors, now that the books have their places
    for (size_t i = 0; i < S.residues.size(); i++) {
      if (vis_lds[i] == 0xFFFFFFFFu) continue;
      const nvh::Residue& r = S.residues[i];
      for (int c = 0; c < r.classifications && c < NVH_MAX_CLASSES; c++) {
        unsigned mask = 0;
        for (int k = 0; k < NVH_MAX_STAGES; k++)
          if (k < r.max_stages && (r.cascade[c] & (1 << k)) && r.books[c][k] >= 0) mask |= 1u << k;
        for (int k = 0; k < NVH_MAX_STAGES; k++) {
          uint32_t* V = &lds_image[vis_lds[i] + 4u * (uint32_t)(c * NVH_MAX_STAGES + k)];
          V[0] = NVH_PVIS_NONE; V[1] = V[2] = V[3] = 0;
          // (the walk tests the cascade bit and the book number, Residue0.cs:160-163; a stage beyond max_stages is never visited)
          if (!(r.cascade[c] & (1 << k)) || r.books[c][k] < 0) continue;
          V[0] = NVH_PVIS_SLOW;
          const int b = r.books[c][k];
          if (b >= (int)S.books.size() || k >= r.max_stages) continue;
          const NvhPBook& d = books[(size_t)b];
          const uint32_t dims = d.dims;
          if (r.type == 0 || !d.has_tree || d.lds_off == 0xFFFFFFFFu || d.lds_off > 0xFFFFFFu || d.prefix_bits < 1 || d.prefix_bits > 24 ||
              dims < 1 || dims > 31 || b > 0xFFFF || r.partition_size < 1)
            continue;
          const uint32_t slots = ((uint32_t)r.partition_size + dims - 1) / dims;
          const uint32_t lat_off = d.slab_lat & 0xFFFFu, lat_values = d.slab_lat >> 16;
          if (slots > 0xFFFFu || lat_off > NVH_SLAB_MAX_LAT_OFF || lat_values > 0xFFu || d.slab_dm16 > 0xFFFFu) continue;
          const uint32_t rank = (uint32_t)__builtin_popcount(mask & ((1u << k) - 1u));
          const uint32_t rw[2] = {NVH_SLAB_REC(0u, d.slab_dm16, lat_off, lat_values, dims, 0u, (uint32_t)k, (mask >> (k + 1)) != 0)};
          V[0] = d.lds_off | ((uint32_t)d.prefix_bits << 24);
          V[1] = slots | (dims << 16) | (rank << 24);
          V[2] = (rw[0] & 0xFFFF0000u) | (uint32_t)b;
          V[3] = rw[1];
        }
      }
    }
  }
  std::vector<NvhPFloor1> floors(S.floors.size());
  for (size_t i = 0; i < S.floors.size(); i++) {
    const nvh::Floor1& f = S.floors[i].f1;
    NvhPFloor1& d = floors[i];
    std::memset(&d, 0, sizeof d);
    d.type = 1;
    d.partition_count = f.partition_count;
    d.y_bits = f.y_bits;
    for (int k = 0; k < 32; k++) d.partition_class[k] = (uint8_t)f.partition_class[k];
    for (int k = 0; k < 16; k++) {
      d.class_dims[k] = (uint8_t)f.class_dimensions[k];
      d.class_sub_bits[k] = (uint8_t)f.class_subclasses[k];
      d.class_master[k] = (int16_t)f.class_masterbook[k];
      for (int j = 0; j < 8; j++) d.sub_book[k][j] = (int16_t)f.subclass_book[k][j];
    }
  }
  std::vector<int32_t> ipool;
  std::vector<NvhPResidue> residues(S.residues.size());
  std::vector<int> r_parts(S.residues.size()), r_ops(S.residues.size()), r_ent(S.residues.size());
  int cap_parts = 1;
  for (size_t i = 0; i < S.residues.size(); i++) {
    const nvh::Residue& r = S.residues[i];
    NvhPResidue& d = residues[i];
    std::memset(&d, 0, sizeof d);
    d.type = r.type; d.begin = r.begin; d.end = r.end; d.partition_size = r.partition_size;
    d.classifications = r.classifications; d.class_book = r.class_book; d.channels = r.channels;
    d.real_channels = r.real_channels; d.max_stages = r.max_stages; d.partvals = r.partvals;
    d.class_dims = S.books[(size_t)r.class_book].dimensions;
    d.alias_b1 = (i < sh.slab.residue_b1.size() && sh.slab.residue_b1[i]) ? 1 : 0;
    d.general = (i < sh.slab.residue_general.size() && sh.slab.residue_general[i] && !d.alias_b1 &&
                 !(i < sh.slab.residue_pair.size() && sh.slab.residue_pair[i])) ? 1u : 0u;
    d.rch_magic = r.real_channels > 1 ? (uint32_t)((0x100000000ull + (uint64_t)r.real_channels - 1) / (uint64_t)r.real_channels) : 0u;
    if ((uint64_t)S.block1 * (uint64_t)std::max(S.channels, 1) * (uint64_t)std::max(r.real_channels, 1) >= 0x100000000ull) return NVH_OK;
    d.decode_map_off = (uint32_t)ipool.size();
    d.decode_map_lds = dm_lds[i];
    d.vis_lds = vis_lds[i];
    d.span_max = nvh::residue_max_span(S, r);
    ipool.insert(ipool.end(), r.decode_map.begin(), r.decode_map.end());
    int min_dims = 1 << 30;
    for (int c = 0; c < NVH_MAX_CLASSES; c++) {
      d.cascade[c] = (uint8_t)r.cascade[c];
      for (int k = 0; k < NVH_MAX_STAGES; k++) {
        d.books[c][k] = (int16_t)r.books[c][k];
        if (k < r.max_stages && (r.cascade[c] & (1 << k)) && r.books[c][k] >= 0) d.book_mask[c] |= (uint8_t)(1u << k);
        if (c < r.classifications && r.books[c][k] >= 0) {
          const int dm = S.books[(size_t)r.books[c][k]].dimensions;
          // k_parse divides (partition_size + dims - 1) by dims with the book's 32-bit reciprocal: exact below 2^32 / dims
          if (dm > 1 && ((uint64_t)r.partition_size + (uint64_t)dm) * (uint64_t)dm >= 0x100000000ull) return NVH_OK;
          if (dm > 0 && dm < min_dims) min_dims = dm;
        }
      }
    }
    if (min_dims == (1 << 30)) min_dims = 1;
    // worst case over this setup's largest block: every partition of every channel has a book in every stage
    const int bs = r.type == 2 ? S.block1 * r.real_channels : S.block1;
    const int end = r.end < bs / 2 ? r.end : bs / 2;
    const int n = end - r.begin;
    const int parts = (n > 0 && r.partition_size > 0) ? n / r.partition_size : 0;
    const int cdim = d.class_dims > 0 ? d.class_dims : 1;
    const int words = (parts + cdim - 1) / cdim;
    r_parts[i] = parts;
    r_ops[i] = r.max_stages * parts * r.channels;
    r_ent[i] = r_ops[i] * ((r.partition_size + min_dims - 1) / min_dims);
    const int need = r.channels * std::max(std::max(parts, words), 1);
    if (need > cap_parts) cap_parts = need;
  }
  if (ipool.empty()) ipool.push_back(0);
  std::vector<NvhPMapping> mappings(S.mappings.size());
  int cap_ops = 1, cap_ent = 8, cap_pass = 1;
  for (size_t i = 0; i < S.mappings.size(); i++) {
    const nvh::Mapping& m = S.mappings[i];
    NvhPMapping& d = mappings[i];
    std::memset(&d, 0, sizeof d);
    d.submaps = (int32_t)m.submap_floor.size();
    d.coupling_steps = (int32_t)m.coupling_angle.size();
    int ops = 0, ent = 0;
    for (size_t k = 0; k < m.submap_floor.size(); k++) {
      d.submap_floor[k] = (uint8_t)m.submap_floor[k];
      d.submap_residue[k] = (uint8_t)m.submap_residue[k];
      ops += r_ops[(size_t)m.submap_residue[k]];
      ent += r_ent[(size_t)m.submap_residue[k]];
    }
    for (int c = 0; c < S.channels; c++) {
      d.chan_floor[c] = (uint8_t)m.channel_floor[(size_t)c];
      d.chan_residue[c] = (uint8_t)m.channel_residue[(size_t)c];
    }
    for (size_t k = 0; k < m.coupling_angle.size(); k++) {
      d.coupling_ang[k] = (uint8_t)m.coupling_angle[k];
      d.coupling_mag[k] = (uint8_t)m.coupling_magnitude[k];
    }
    if (ops > cap_ops) cap_ops = ops;
    if (ent > cap_ent) cap_ent = ent;
    if (d.submaps > cap_pass) cap_pass = d.submaps;
  }
  cap_ops = (cap_ops + 7) & ~7;
  cap_ent = (cap_ent + 15) & ~7;
  // keep a frame's slabs within reason (and op indices within the 15-bit links where possible)
  if ((size_t)cap_ops * 10 + (size_t)cap_ent * 2 + (size_t)cap_parts * 8 > ((size_t)1 << 20)) return NVH_OK;

  ArenaBuilder ab;
  // books | floors | residues | mappings back to back: k_parse copies this block into LDS
  size_t o_bk = ab.add(books.data(), books.size() * sizeof(NvhPBook));
  size_t o_fl = ab.add(floors.data(), floors.size() * sizeof(NvhPFloor1));
  size_t o_rs = ab.add(residues.data(), residues.size() * sizeof(NvhPResidue));
  size_t o_mp = ab.add(mappings.data(), mappings.size() * sizeof(NvhPMapping));
  const size_t meta_end = (ab.bytes.size() + 15) / 16 * 16;
  size_t o_px = ab.add(prefix.data(), prefix.size() * sizeof(uint32_t));
  NvhPOverflow none{};
  size_t o_ov = ab.add(overflow.empty() ? &none : overflow.data(), (overflow.empty() ? 1 : overflow.size()) * sizeof(NvhPOverflow));
  size_t o_ip = ab.add(ipool.data(), ipool.size() * sizeof(int32_t));
  size_t o_li = ab.add(lds_image.data(), lds_image.size() * sizeof(uint32_t));
  size_t o_si = ab.add(sub_image.data(), sub_image.size() * sizeof(uint32_t));
  if (meta_end - o_bk > 8 * 1024) return NVH_OK;  // unusually large setup: keep the host parser
  sh.parse_arena.pool = &s->ctx->pool;
  int rc = sh.parse_arena.reserve(ab.bytes.size());
  if (rc != NVH_OK) return rc;
  HIP_TRY(hipMemcpy(sh.parse_arena.p, ab.bytes.data(), ab.bytes.size(), hipMemcpyHostToDevice));
  const uint8_t* base = (const uint8_t*)sh.parse_arena.p;
  NvhDevParse& P = sh.parse;
  P.channels = S.channels;
  P.block1 = S.block1;
  P.cap_pass = cap_pass;
  P.cap_ops = cap_ops;
  P.cap_ent = cap_ent;
  P.cap_parts = cap_parts;
  P.books = (const NvhPBook*)(base + o_bk);
  P.prefix = (const uint32_t*)(base + o_px);
  P.overflow = (const NvhPOverflow*)(base + o_ov);
  P.floors = (const NvhPFloor1*)(base + o_fl);
  P.residues = (const NvhPResidue*)(base + o_rs);
  P.mappings = (const NvhPMapping*)(base + o_mp);
  P.ipool = (const int32_t*)(base + o_ip);
  P.lds_image = (const uint32_t*)(base + o_li);
  P.lds_words = (int32_t)lds_image.size();
  P.sub_image = (const uint32_t*)(base + o_si);
  P.sub_words = (int32_t)sub_image.size();
  P.meta_words = (int32_t)((meta_end - o_bk) / 4);
  P.meta_floors_off = (int32_t)(o_fl - o_bk);
  P.meta_residues_off = (int32_t)(o_rs - o_bk);
  P.meta_mappings_off = (int32_t)(o_mp - o_bk);
  P.dm_in_lds = 1;
  for (uint32_t o : dm_lds) if (o == 0xFFFFFFFFu) P.dm_in_lds = 0;
  for (uint32_t o : vis_lds) if (o == 0xFFFFFFFFu) P.dm_in_lds = 0;
  P.slab_general = 0;
  P.row_words = 2 * cap_parts;
  // slab mode: setups inside the slab kernels' contract whose frames have one residue pass, lattice offsets and values a record
  // can hold (+ room for the partition table of a quirk-B-1 residue)
  P.dfloors = sh.dev.floors;
  P.recip = sh.dev.recip;
  P.max_posts = sh.max_posts;
  P.slab_stride_vecs = 0;
  {
    // (streams of the general bin walk -- several passes per frame, Residue0, odd dimensions, aliasing stereo Residue2: round 5 --
    // carry a group list: the count, two units per (pass, channel), one chain index per (pass, channel, partition))
    const bool general = sh.slab_general;
    bool ok = sh.slab_setup_ok && (general ? (cap_pass <= 15 && !s->has_floor0) : cap_pass <= 1) && S.channels <= NVH_SLAB_MAX_CH;
    for (const NvhDevBook& db : sh.slab.books) ok = ok && db.lat_off <= NVH_SLAB_MAX_LAT_OFF && db.lat_values <= 0xFFu;
    const size_t Pn = (size_t)sh.max_posts + 2;
    size_t v = NVH_SLAB_HDR_VECS + (size_t)S.channels * (Pn + ((size_t)S.block1 / 8 + 15) / 16) + ((size_t)cap_ops + 3) / 4 + ((size_t)cap_ops + 1) / 2 +
               ((size_t)cap_ent + 7) / 8 + 1 + 1 + ((size_t)cap_parts + 7) / 8;
    if (general) {
      v += 1 + 2 * (size_t)cap_pass * (size_t)S.channels + ((size_t)cap_pass * (size_t)cap_parts + 7) / 8;
      P.slab_general = 1;
      P.row_words = cap_parts + cap_pass * (cap_parts + 2);
    }
    if (v < (size_t)S.block1 / 64 + 8) v = (size_t)S.block1 / 64 + 8;
    v = (v + 3) & ~(size_t)3;
    if (ok && v <= 0xFFFFu && cap_ops <= 0xFFFF && cap_ent <= 0xFFFF) P.slab_stride_vecs = (int32_t)v;
  }
  sh.gpu_parse_ok = true;
  return NVH_OK;
}
