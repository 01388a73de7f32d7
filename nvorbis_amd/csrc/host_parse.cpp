// host_parse.cpp -- audio-packet bit parser + stream geometry (product host code).
//
// Reference behaviour followed (file:line under /root/reference/NVorbis/):
//   StreamDecoder.cs:320-530 (Read / ReadNextPacket / DecodeNextPacket, integer state only),
//   Mode.cs:119-151 (GetPacketInfo), Mapping.cs:95-134 (bit-consuming half of DecodePacket),
//   Floor1.cs:135-184 (Unpack), Floor0.cs:98-150 (Unpack), Residue0.cs:119-201, Residue1.cs:8-26,
//   Residue2.cs:16-47 (classification + entry decode; the float adds are replayed on the GPU).
#include "host_parse.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace nvh {

void PacketPool::release() {
  if (base) {
    if (grow) (void)grow(owner, base, cap, 0);
    else std::free(base);
  }
  base = nullptr;
  size = cap = 0;
}

uint8_t* PacketPool::append(size_t n) {
  if (size + n > cap) {
    size_t want = cap ? cap * 2 : (size_t)1 << 16;
    while (want < size + n) want *= 2;
    uint8_t* nb = grow ? grow(owner, nullptr, 0, want) : static_cast<uint8_t*>(std::malloc(want));
    if (!nb) return nullptr;
    if (size) std::memcpy(nb, base, size);
    if (base) {
      if (grow) (void)grow(owner, base, cap, 0);
      else std::free(base);
    }
    base = nb;
    cap = want;
  }
  uint8_t* p = base + size;
  size += n;
  return p;
}

// ---------------------------------------------------------------------------------------------
// floors
// ---------------------------------------------------------------------------------------------

// Codebook.DecodeScalar (Codebook.cs:294-320) with the common case in line: eight bytes of packet behind the cursor's byte and a
// code the prefix table resolves -- then the peek returns every bit it was asked for and the skip stays inside the packet, which is
// all the reference's end-of-packet rules are about; everything else goes through Codebook::decode_scalar.
static inline int decode_scalar_fast(const Codebook& book, BitReader& p) {
  if (!book.fast.empty() && (p.pos >> 3) + 8 <= (p.total_bits >> 3)) {
    uint64_t w;
    std::memcpy(&w, p.data + (p.pos >> 3), 8);
    const uint32_t node = book.fast[(uint32_t)(w >> (p.pos & 7)) & ((1u << book.prefix_bits) - 1u)];
    if (node & 0x80u) {
      p.pos += (int)(node & 0x7Fu);
      return (int)(node >> 8);
    }
  }
  return book.decode_scalar(p);
}

int StreamParser::decode_floor(int floor_idx, BitReader& p, FrameBatch& out, NvhChan& ch, bool* energy) {
  const Floor& fl = s_->floors[(size_t)floor_idx];
  ch.floor = (uint8_t)floor_idx;
  ch.post_count = 0;
  ch.amp = 0.0f;
  ch.data_off = 0;
  *energy = false;
  if (fl.type == 1) {
    // Floor1.Unpack (Floor1.cs:135-184)
    const Floor1& f = fl.f1;
    int posts[NVH_MAX_POSTS];
    int post_count = 0;
    if (p.read_bit()) {
      post_count = 2;
      posts[0] = (int)p.read(f.y_bits);
      posts[1] = (int)p.read(f.y_bits);
      for (int i = 0; i < f.partition_count; i++) {
        int cls = f.partition_class[i];
        int cdim = f.class_dimensions[cls];
        int cbits = f.class_subclasses[cls];
        int csub = (1 << cbits) - 1;
        uint32_t cval = 0;
        if (cbits > 0) {
          int r = decode_scalar_fast(s_->books[(size_t)f.class_masterbook[cls]], p);
          if (r == -2) return NVH_ERR_RUNTIME;
          cval = (uint32_t)r;
          if (cval == 0xFFFFFFFFu) {
            post_count = 0;
            break;
          }
        }
        for (int j = 0; j < cdim; j++) {
          int book = f.subclass_book[cls][cval & (uint32_t)csub];
          cval >>= cbits;
          if (book >= 0) {
            if (post_count >= NVH_MAX_POSTS) return NVH_ERR_RUNTIME;  // Posts = new int[64]
            int r = decode_scalar_fast(s_->books[(size_t)book], p);
            if (r == -2) return NVH_ERR_RUNTIME;
            if ((posts[post_count] = r) == -1) {
              post_count = 0;
              i = f.partition_count;
              break;
            }
          } else if (post_count < NVH_MAX_POSTS) {
            posts[post_count] = 0;  // Posts[] is zero-initialised and never written for a null book
          }
          ++post_count;
        }
      }
    }
    if (post_count > NVH_MAX_POSTS) return NVH_ERR_RUNTIME;  // UnwrapPosts would index past finalY[64]
    ch.post_count = (uint8_t)post_count;
    ch.data_off = (uint32_t)out.posts.size();
    for (int i = 0; i < post_count; i++) {
      if (posts[i] < 0 || posts[i] > 0xFFFF) return NVH_ERR_UNSUPPORTED;  // documented limit: raw post values fit 16 bits
      out.posts.push_back((uint16_t)posts[i]);
    }
    *energy = post_count > 0;
    return NVH_OK;
  }

  // Floor0.Unpack (Floor0.cs:98-150)
  const Floor0& f = fl.f0;
  std::vector<float> coeff((size_t)f.order + 1, 0.0f);
  float amp = (float)p.read(f.amp_bits);
  if (amp > 0.0f) {
    amp = amp / (float)f.amp_div * (float)f.amp_ofs;
    uint32_t book_num = (uint32_t)p.read(f.book_bits);
    if (book_num >= (uint32_t)f.books.size()) {
      amp = 0.0f;
    } else {
      const Codebook& book = s_->books[(size_t)f.books[book_num]];
      bool ok = true;
      for (int i = 0; i < f.order && ok;) {
        int entry = book.decode_scalar(p);
        if (entry == -2) return NVH_ERR_RUNTIME;
        if (entry == -1) {
          amp = 0.0f;
          ok = false;
          break;
        }
        for (int j = 0; i < f.order && j < book.dimensions; j++, i++) coeff[i] = book.lookup[(size_t)entry * book.dimensions + j];
      }
      if (ok) {
        float last = 0.0f;
        for (int j = 0; j < f.order;) {
          for (int k = 0; j < f.order && k < book.dimensions; j++, k++) coeff[j] += last;
          last = coeff[j - 1];
        }
      }
    }
  }
  ch.amp = amp;
  ch.post_count = amp > 0.0f ? 1 : 0;
  ch.data_off = (uint32_t)out.coeffs.size();
  out.coeffs.insert(out.coeffs.end(), coeff.begin(), coeff.end());
  *energy = amp > 0.0f;
  return NVH_OK;
}

// ---------------------------------------------------------------------------------------------
// residues
// ---------------------------------------------------------------------------------------------

int StreamParser::decode_residue(int residue_idx, BitReader& p, int block_size, FrameBatch& out, NvhResPass& pass,
                                 uint32_t frame_op_begin) {
  const Residue& r = s_->residues[(size_t)residue_idx];
  const Codebook& class_book = s_->books[(size_t)r.class_book];
  pass.residue = residue_idx;
  for (int s = 0; s <= NVH_MAX_STAGES; s++) pass.op_begin[s] = (uint32_t)out.ops.size();

  if (r.type == 2) block_size *= r.real_channels;  // Residue2.cs:16-21
  int end = r.end < block_size / 2 ? r.end : block_size / 2;
  int n = end - r.begin;
  if (n <= 0) return NVH_OK;

  int partition_count = n / r.partition_size;
  int cdim = class_book.dimensions;
  if (cdim == 0) return NVH_ERR_RUNTIME;
  int partition_words = (partition_count + cdim - 1) / cdim;
  std::vector<int>& part_word = scratch_part_word_;  // (members: no allocation per packet)
  part_word.assign((size_t)r.channels * (size_t)std::max(partition_words, 1), -1);
  const int buflen = s_->block1;  // float[ch][block1Size] (StreamDecoder.cs:498-505)
  bool stop = false;
  // op_link chains: the last op emitted for each (partition, channel) of this pass
  std::vector<int32_t>& last_op = scratch_last_op_;
  last_op.assign((size_t)r.channels * (size_t)std::max(partition_count, 1), -1);
  auto link_op = [&](int partition_idx, int ch) {
    const size_t idx = out.ops.size() - 1;
    out.op_link.resize(out.ops.size(), (uint16_t)NVH_LINK_NONE);
    out.op_link[idx] = (uint16_t)NVH_LINK_NONE;
    const size_t rel = idx - (size_t)frame_op_begin;
    if (rel >= (size_t)NVH_LINK_NONE) out.links_ok = false;
    int32_t& last = last_op[(size_t)partition_idx * (size_t)r.channels + (size_t)ch];
    if (last >= 0 && rel < (size_t)NVH_LINK_NONE) {
      out.op_link[(size_t)last] = (uint16_t)((out.op_link[(size_t)last] & 0x8000u) | (uint16_t)rel);
      out.op_link[idx] |= 0x8000u;
    }
    last = (int32_t)idx;
  };

  int stage = 0;
  for (; stage < r.max_stages && !stop; stage++) {
    pass.op_begin[stage] = (uint32_t)out.ops.size();
    for (int partition_idx = 0, entry_idx = 0; partition_idx < partition_count && !stop; entry_idx++) {
      if (stage == 0) {
        for (int ch = 0; ch < r.channels; ch++) {
          int idx = decode_scalar_fast(class_book, p);
          if (idx == -2) return NVH_ERR_RUNTIME;
          if (idx >= 0 && idx < r.partvals) {
            part_word[(size_t)ch * partition_words + entry_idx] = idx;
          } else {
            stop = true;
            break;
          }
        }
        if (stop) break;
      }
      for (int dimension_idx = 0; partition_idx < partition_count && dimension_idx < cdim && !stop;
           dimension_idx++, partition_idx++) {
        int offset = r.begin + partition_idx * r.partition_size;
        for (int ch = 0; ch < r.channels; ch++) {
          int word = part_word[(size_t)ch * partition_words + entry_idx];
          if (word < 0) return NVH_ERR_RUNTIME;  // NullReferenceException on partWordCache
          int cls = r.decode_map[(size_t)word * cdim + dimension_idx];
          if ((r.cascade[cls] & (1 << stage)) == 0) continue;
          int book_idx = r.books[cls][stage];
          if (book_idx < 0) continue;
          const Codebook& book = s_->books[(size_t)book_idx];
          int dims = book.dimensions;
          if (dims == 0) return NVH_ERR_RUNTIME;

          NvhResOp op;
          op.ent_off = (uint32_t)out.entries.size();
          op.partition = (uint16_t)partition_idx;
          op.channel = (uint8_t)ch;
          op.book = (uint8_t)book_idx;
          if (partition_idx > 0xFFFF) return NVH_ERR_UNSUPPORTED;

          if (r.type == 0) {
            // Residue0.WriteVectors (:180-201): decode all entries first, add only if all decoded
            int steps = r.partition_size / dims;
            size_t mark = out.entries.size();
            bool bad = false;
            for (int i = 0; i < steps; i++) {
              int e = book.decode_scalar(p);
              if (e == -2) return NVH_ERR_RUNTIME;
              if (e == -1) {
                bad = true;
                break;
              }
              out.entries.push_back((uint16_t)e);
            }
            if (bad) {
              out.entries.resize(mark);
              stop = true;
              break;
            }
            if (offset + steps * dims > buflen) return NVH_ERR_RUNTIME;
            out.ops.push_back(op);
            link_op(partition_idx, ch);
          } else {
            // Residue1.WriteVectors (Residue1.cs:8-26) / Residue2.WriteVectors (Residue2.cs:23-47):
            // vectors are added as they are decoded; a failed decode keeps what was added so far
            int slots = (r.partition_size + dims - 1) / dims;
            int done = 0;
            bool bad = false;
            const size_t ebase = out.entries.size();
            out.entries.resize(ebase + (size_t)slots);
            uint16_t* eo = out.entries.data() + ebase;
            // The vector's entries, fast form (the GPU parser's loop, kernels_parse.hip): while eight bytes of packet lie behind
            // the cursor's byte and the prefix table resolves the code, a symbol is an unaligned load, a shift, a table read
            // and a store.  Codebook.DecodeScalar does exactly that in this situation (the peek returns every bit asked for,
            // the skip stays inside the packet); whatever the loop leaves -- a long code, the packet's last bytes -- goes
            // through decode_scalar with the reference's end-of-packet rules.
            if (book.has_tree && !book.fast.empty()) {
              const uint32_t* ft = book.fast.data();
              const uint32_t mask = (1u << book.prefix_bits) - 1u;
              const uint8_t* bytes = p.data;
              const int nbytes = p.total_bits >> 3;
              int pos = p.pos;
              while (done < slots && (pos >> 3) + 8 <= nbytes) {
                uint64_t w;
                std::memcpy(&w, bytes + (pos >> 3), 8);
                const uint32_t node = ft[(uint32_t)(w >> (pos & 7)) & mask];
                if (!(node & 0x80u)) break;
                pos += (int)(node & 0x7Fu);
                eo[done++] = (uint16_t)(node >> 8);
              }
              p.pos = pos;
            }
            for (int i = done * dims; i < r.partition_size; i += dims) {
              int e = book.decode_scalar(p);
              if (e == -2) return NVH_ERR_RUNTIME;
              if (e == -1) {
                bad = true;
                break;
              }
              eo[done++] = (uint16_t)e;
            }
            // bounds of the adds the reference performed
            if (done > 0) {
              int last = done * dims - 1;
              if (r.type == 1) {
                if (offset + last >= buflen) return NVH_ERR_RUNTIME;
              } else {
                if (offset / r.real_channels + last / r.real_channels >= buflen) return NVH_ERR_RUNTIME;
              }
            }
            for (int i = done; i < slots; i++) eo[i] = (uint16_t)NVH_ENTRY_SKIP;
            out.ops.push_back(op);
            link_op(partition_idx, ch);
            if (bad) {
              stop = true;
              break;
            }
          }
        }
      }
    }
  }
  // stages not reached keep empty ranges
  uint32_t end_ops = (uint32_t)out.ops.size();
  int first_unset = stop ? stage : stage;  // `stage` is one past the last stage that ran
  for (int s = first_unset; s <= NVH_MAX_STAGES; s++) pass.op_begin[s] = end_ops;
  return NVH_OK;
}

// ---------------------------------------------------------------------------------------------
// one audio packet
// ---------------------------------------------------------------------------------------------

int StreamParser::parse_audio(BitReader& p, FrameBatch& out, int* decoded) {
  *decoded = 0;
  int mode_idx = (int)p.read(s_->mode_field_bits);
  if (mode_idx >= (int)s_->modes.size()) return NVH_ERR_RUNTIME;  // _modes[...] out of range (quirk B-15)
  const Mode& m = s_->modes[(size_t)mode_idx];

  // Mode.GetPacketInfo (Mode.cs:119-151)
  if (p.is_short) return NVH_OK;
  int wi = 0, start, valid, total;
  if (m.block_flag) {
    bool prev_flag = p.read_bit();
    bool next_flag = p.read_bit();
    wi = (prev_flag ? 1 : 0) + (next_flag ? 2 : 0);
    start = m.ov_start[wi];
    valid = m.ov_valid[wi];
    total = m.ov_total[wi];
  } else {
    start = 0;
    valid = m.block_size / 2;
    total = m.block_size;
  }

  // Mapping.DecodePacket, bit-consuming half (Mapping.cs:95-134)
  const Mapping& map = s_->mappings[(size_t)m.mapping];
  const int nch = s_->channels;
  const int bs = m.block_size;

  // roll back everything appended by this packet if the reference would have thrown
  NvhFrame f;
  std::memset(&f, 0, sizeof f);
  f.n = bs;
  f.mapping = m.mapping;
  f.window_off = m.window_off[wi];
  f.mdct_slot = m.block_flag ? 1 : 0;
  f.start = start;
  f.valid = valid;
  f.total = total;
  f.ov_frame = -1;
  f.chan_off = (uint32_t)out.chans.size();
  f.pass_begin = (uint32_t)out.passes.size();
  f.op_begin = (uint32_t)out.ops.size();
  f.ent_begin = (uint32_t)out.entries.size();

  if (light_) {
    // GPU-parse mode: the rest of the packet is decoded by kernels_parse.hip.  Keep the packet (word aligned, zero
    // padded: the device bit reader loads whole words) and where the device should continue reading.
    if (nch > NVH_PARSE_MAX_CH) return NVH_ERR_UNSUPPORTED;
    out.pkt_refs.resize(out.frames.size());  // pseudo-frames carry empty references
    NvhPacketRef ref;
    ref.byte_off = (uint32_t)out.pkt_pool.size;
    ref.bit_len = (uint32_t)p.total_bits;
    ref.bit_pos = (uint32_t)p.pos;
    ref.pad = 0;
    const size_t nbytes = (size_t)(p.total_bits >> 3);
    const size_t padded = ((out.pkt_pool.size + nbytes + 7) & ~(size_t)3) - out.pkt_pool.size;  // >= 4 bytes of zeros behind every packet
    uint8_t* dst = out.pkt_pool.append(padded);
    if (!dst) return NVH_ERR_NOMEM;
    if (nbytes) std::memcpy(dst, p.data, nbytes);
    std::memset(dst + nbytes, 0, padded - nbytes);
    out.pkt_refs.push_back(ref);
    for (int i = 0; i < nch; i++) {
      NvhChan ch;
      std::memset(&ch, 0, sizeof ch);
      out.chans.push_back(ch);
    }
    f.pass_end = f.pass_begin;
    out.frames.push_back(f);
    *decoded = 1;
    return NVH_OK;
  }
  std::vector<uint8_t> energy((size_t)nch), force_energy((size_t)nch, 0), force_no_energy((size_t)nch, 0);
  bool any_execute = false;
  for (int i = 0; i < nch; i++) {
    NvhChan ch;
    std::memset(&ch, 0, sizeof ch);
    bool e = false;
    int rc = decode_floor(map.channel_floor[(size_t)i], p, out, ch, &e);
    if (rc != NVH_OK) return rc;
    energy[i] = e;
    any_execute |= e;  // noExecuteChannel[i] = !ExecuteChannel, computed before ForceEnergy (quirk B-5)
    out.chans.push_back(ch);
  }
  auto exec = [&](int c) { return (force_energy[c] || energy[c]) && !force_no_energy[c]; };

  for (size_t i = 0; i < map.coupling_angle.size(); i++) {  // Mapping.cs:112-119
    int a = map.coupling_angle[i], mg = map.coupling_magnitude[i];
    if (exec(a) || exec(mg)) {
      force_energy[a] = 1;
      force_energy[mg] = 1;
    }
  }
  for (size_t i = 0; i < map.submap_floor.size(); i++) {  // Mapping.cs:122-134
    for (int j = 0; j < nch; j++) {
      if (map.submap_floor[i] != map.channel_floor[(size_t)j] || map.submap_residue[i] != map.channel_residue[(size_t)j])
        force_no_energy[j] = 1;
    }
    if (any_execute) {  // Array.IndexOf(doNotDecodeChannel, false) != -1 (Residue0.cs:125)
      NvhResPass pass;
      int rc = decode_residue(map.submap_residue[i], p, bs, out, pass, f.op_begin);
      if (rc != NVH_OK) return rc;
      out.passes.push_back(pass);
    }
  }
  for (int c = 0; c < nch; c++) {
    out.chans[f.chan_off + (size_t)c].exec = exec(c) ? 1 : 0;
    if (c < 32 && exec(c)) f.exec_mask |= 1u << c;
  }
  f.pass_end = (uint32_t)out.passes.size();
  f.op_count = (uint32_t)out.ops.size() - f.op_begin;
  f.ent_count = (uint32_t)out.entries.size() - f.ent_begin;
  out.frames.push_back(f);
  *decoded = 1;
  return NVH_OK;
}

int StreamParser::parse_residue(int residue_idx, const uint8_t* data, int len, int bit_offset, int block_size, FrameBatch& out,
                                int* bits_consumed) {
  if (residue_idx < 0 || residue_idx >= (int)s_->residues.size()) return NVH_ERR_ARGUMENT;
  BitReader p(data, len);
  p.skip(bit_offset);
  NvhFrame f;
  std::memset(&f, 0, sizeof f);
  f.n = block_size;
  f.mdct_slot = block_size == s_->block1 ? 1 : 0;
  f.chan_off = (uint32_t)out.chans.size();
  f.pass_begin = (uint32_t)out.passes.size();
  f.op_begin = (uint32_t)out.ops.size();
  f.ent_begin = (uint32_t)out.entries.size();
  NvhChan ch;
  std::memset(&ch, 0, sizeof ch);
  for (int c = 0; c < s_->channels; c++) out.chans.push_back(ch);
  NvhResPass pass;
  int rc = decode_residue(residue_idx, p, block_size, out, pass, f.op_begin);
  if (rc != NVH_OK) return rc;
  out.passes.push_back(pass);
  f.pass_end = (uint32_t)out.passes.size();
  f.op_count = (uint32_t)out.ops.size() - f.op_begin;
  f.ent_count = (uint32_t)out.entries.size() - f.ent_begin;
  out.frames.push_back(f);
  if (bits_consumed) *bits_consumed = p.pos - bit_offset;
  return NVH_OK;
}

// ---------------------------------------------------------------------------------------------
// stream state machine
// ---------------------------------------------------------------------------------------------

void StreamParser::drain(FrameBatch& out) {
  // `_prevPacketEnd = _prevPacketStop` (StreamDecoder.cs:352-356): the previous block's windowed tail
  // is emitted as it is, with nothing overlapped onto it.
  int cnt = prev_stop_ - prev_start_;
  if (cnt > 0) {
    if (prev_frame_ >= 0) {
      out.frames[(size_t)prev_frame_].emit_count += cnt;
    } else if (prev_frame_ == -2) {
      NvhFrame f;
      std::memset(&f, 0, sizeof f);
      f.n = 0;  // no synthesis: emits the carried tail
      f.ov_frame = -2;
      f.ov_src = prev_start_;
      f.ov_len = cnt;
      f.ov_n = prev_n_;
      f.ov_window_off = prev_window_off_;
      f.emit_start = 0;
      f.emit_count = cnt;
      f.out_pos = out.pcm_samples;
      f.chan_off = (uint32_t)out.chans.size();
      f.pass_begin = f.pass_end = (uint32_t)out.passes.size();
      for (size_t c = 0; c < (size_t)s_->channels; c++) {
        NvhChan ch;
        std::memset(&ch, 0, sizeof ch);
        ch.ov_exec = c < prev_exec_.size() ? prev_exec_[c] : 0;
        if (c < 32 && ch.ov_exec) f.ov_exec_mask |= 1u << c;
        out.chans.push_back(ch);  // the pseudo-frame carries the source frame's exec flags in its own channel records
      }
      out.frames.push_back(f);
    }
    out.pcm_samples += cnt;
    position_ += cnt;
    emitted_ += cnt;
  }
  prev_end_ = prev_stop_;
  prev_start_ = prev_end_;
}

int StreamParser::push_end(FrameBatch& out) {
  if (eos_found_) return NVH_OK;
  eos_found_ = true;  // GetNextPacket() == null (StreamDecoder.cs:472-475)
  drain(out);
  return NVH_OK;
}

void StreamParser::begin_batch() {
  if (prev_frame_ >= 0) prev_frame_ = -2;
}

int StreamParser::push_packet(const uint8_t* data, int len, int64_t granule, int flags, FrameBatch& out) {
  if (eos_found_) return NVH_OK;  // Read() stops pulling packets once _eosFound (StreamDecoder.cs:343-350)
  BitReader p(data, len);
  const bool is_eos = (flags & NVH_PKT_EOS) != 0;
  if (flags & NVH_PKT_RESYNC) has_position_ = false;  // StreamDecoder.cs:481-484

  // transactional append: a packet that makes the reference throw leaves the batch untouched
  const size_t m_frames = out.frames.size(), m_chans = out.chans.size(), m_passes = out.passes.size(),
               m_ops = out.ops.size(), m_entries = out.entries.size(), m_posts = out.posts.size(),
               m_coeffs = out.coeffs.size(), m_pool = out.pkt_pool.size, m_refs = out.pkt_refs.size();
  auto rollback = [&]() {
    out.frames.resize(m_frames); out.chans.resize(m_chans); out.passes.resize(m_passes); out.ops.resize(m_ops); out.op_link.resize(m_ops);
    out.entries.resize(m_entries); out.posts.resize(m_posts); out.coeffs.resize(m_coeffs);
    out.pkt_pool.size = m_pool; out.pkt_refs.resize(m_refs);
  };

  int decoded = 0;
  if (!p.read_bit()) {  // StreamDecoder.cs:490
    int rc = parse_audio(p, out, &decoded);
    if (rc != NVH_OK) {
      rollback();
      return rc;
    }
    if (!decoded) rollback();  // floor/residue data of a rejected packet is never used
  }
  eos_found_ = eos_found_ || is_eos;
  if (!decoded) {
    drain(out);
    return NVH_OK;
  }

  // ReadNextPacket (StreamDecoder.cs:417-463)
  const int idx = (int)out.frames.size() - 1;
  NvhFrame& f = out.frames[(size_t)idx];
  int start = f.start, valid = f.valid, total = f.total;
  bool trimmed_eos = false;
  if (granule >= 0 && is_eos) {
    int64_t actual_end = position_ + valid - start;
    int diff = (int)(granule - actual_end);
    if (diff < 0) {
      valid += diff;
      trimmed_eos = true;
    }
  }
  if (prev_end_ > 0) {
    int ov_len = prev_stop_ - prev_start_;
    if (ov_len > 0) {
      if (start + ov_len > s_->block1 || prev_stop_ > s_->block1) {
        rollback();
        return NVH_ERR_RUNTIME;  // IndexOutOfRangeException in OverlapBuffers
      }
      f.ov_frame = prev_frame_;
      f.ov_src = prev_start_;
      f.ov_len = ov_len;
      f.ov_n = prev_n_;
      f.ov_window_off = prev_window_off_;
      for (size_t c = 0; c < prev_exec_.size() && c < (size_t)s_->channels; c++) {
        out.chans[f.chan_off + c].ov_exec = prev_exec_[c];
        if (c < 32 && prev_exec_[c]) f.ov_exec_mask |= 1u << c;
      }
      // The overlap reaches this block's own tail: a later overlap (or drain) would read samples this one has
      // modified, so the batch must apply its overlaps in order.  Measured against the untrimmed `valid`: an
      // end-of-stream trim moves `valid` into the overlapped region, but nothing is pulled after that packet
      // (StreamDecoder.cs:343-350), so its tail is never read.
      if (start + ov_len > (trimmed_eos ? f.valid : valid)) out.sequential_ola = true;
    }
    prev_start_ = start;
  } else if (!has_prev_buf_) {
    prev_start_ = valid;  // first packet: nothing before `valid` is good (StreamDecoder.cs:446-450)
  }
  prev_end_ = valid;
  prev_stop_ = total;
  has_prev_buf_ = true;

  if (granule >= 0 && !has_position_) {  // StreamDecoder.cs:359-363
    has_position_ = true;
    position_ = granule - (prev_end_ - prev_start_);
  }

  int cnt = prev_end_ - prev_start_;
  if (cnt < 0) {
    // the managed Read() loop would never terminate (copyLen < 0, start != end); refuse instead
    rollback();
    return NVH_ERR_RUNTIME;
  }
  f.valid = valid;
  f.emit_start = prev_start_;
  f.emit_count = cnt;
  f.out_pos = out.pcm_samples;
  out.pcm_samples += cnt;
  position_ += cnt;
  emitted_ += cnt;
  prev_start_ = prev_end_;
  prev_frame_ = idx;
  prev_n_ = f.n;
  prev_window_off_ = f.window_off;
  prev_exec_.resize((size_t)s_->channels);
  for (int c = 0; c < s_->channels; c++) prev_exec_[(size_t)c] = out.chans[f.chan_off + (size_t)c].exec;  // light mode: fixed up on the device (k_parse_links)
  return NVH_OK;
}

}  // namespace nvh
