// host_ogg.cpp -- minimal forward-only Ogg demux (product host code).
//
// The container is outside the accelerated path; this exists so that .ogg files can feed it.  It
// reproduces what NVorbis' seekable reader delivers to StreamDecoder for one logical stream of the file:
//   page sync + CRC-32 (poly 0x04c11db7)                      Ogg/PageReaderBase.cs:33-70, Ogg/Crc.cs:5-40
//   lacing -> packets; zero-length packets are dropped         Ogg/PageReader.cs:27-93
//   a page without packets is rejected and blacklists the serial  Ogg/PageReader.cs:131, Ogg/PageReaderBase.cs:72-85
//   continued packets; the granule position goes to the packet that is last on the page it completes
//   on; end-of-stream to that packet of the EOS-flagged page    Ogg/PacketProvider.cs:324-438
//   multiplexed / chained files: pages are routed by serial number to logical streams                Ogg/PageReader.cs:126-158
// Seeking: the page table ogg_demux can return and ogg_seek below restate
//   StreamPageReader.FindPage / FindPageBisection / FindPageForward     Ogg/StreamPageReader.cs:122-264
//   PacketProvider.SeekTo, FindPacket, the libvorbis granule workaround  Ogg/PacketProvider.cs:56-260
//   PacketProvider.NormalizePacketIndex                                   Ogg/PacketProvider.cs:262-295
#if defined(__x86_64__)
#include <cstdlib>
#include <immintrin.h>
#endif
#include <cstdint>
#include <cstring>
#include <vector>

#include "host_ogg.h"
#include "host_setup.h"

namespace nvh {

namespace {

struct Page {
  size_t data_off = 0;
  int flags = 0;
  int64_t granule = 0;
  bool resync = false, continued = false;
  std::vector<int> pk_off, pk_len;
};

// CRC-32 of an Ogg page (polynomial 0x04c11db7, most significant bit first, no reflection, initial value 0: Ogg/Crc.cs:5-40),
// eight bytes per step: t[k][b] = the CRC register after byte b and k zero bytes.
struct CrcTable {
  uint32_t t[8][256];
  CrcTable() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t s = i << 24;
      for (int j = 0; j < 8; ++j) s = (s << 1) ^ (s >= (1u << 31) ? 0x04c11db7u : 0u);
      t[0][i] = s;
    }
    for (int k = 1; k < 8; k++)
      for (uint32_t i = 0; i < 256; i++) t[k][i] = (t[k - 1][i] << 8) ^ t[0][t[k - 1][i] >> 24];
  }
};

inline uint32_t be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

uint32_t crc_table(uint32_t crc, const uint8_t* p, size_t n) {
  static const CrcTable tab;
  while (n >= 8) {
    const uint32_t a = crc ^ be32(p), b = be32(p + 4);
    crc = tab.t[7][a >> 24] ^ tab.t[6][(a >> 16) & 255u] ^ tab.t[5][(a >> 8) & 255u] ^ tab.t[4][a & 255u] ^
          tab.t[3][b >> 24] ^ tab.t[2][(b >> 16) & 255u] ^ tab.t[1][(b >> 8) & 255u] ^ tab.t[0][b & 255u];
    p += 8;
    n -= 8;
  }
  while (n--) crc = (crc << 8) ^ tab.t[0][*p++ ^ (crc >> 24)];
  return crc;
}

#if defined(__x86_64__)
// The same checksum by carry-less multiplication (the corpus pass checks 3.3 GB of pages inside its timed region: a quarter of a
// worker's time at the table's ~1.5 GB/s per core).  A page is a polynomial over GF(2), first byte most significant; 16-byte blocks are
// loaded byte-reversed, so that a register IS that polynomial, and an accumulator a is carried over D bits as
// hi64(a) * (x^(D+64) mod P) + lo64(a) * (x^D mod P) -- congruent to a * x^D modulo P, and short enough for 128 bits (64 x 32-bit
// products).  Four accumulators 64 bytes apart (D = 512), folded into one (D = 128); what is left -- 16 bytes of accumulator, then
// the buffer's last bytes -- goes through the table, which multiplies by x^32 and reduces.  x^128, x^192, x^512, x^576 mod
// 0x104c11db7 = 0xe8a45605, 0xc5b9cd4c, 0xe6228b11, 0x8833794c.  tests/test_host_logic.py compares both forms on random buffers.
#define NVH_CLMUL_TARGET __attribute__((target("pclmul,ssse3")))
NVH_CLMUL_TARGET inline __m128i crc_ld(const uint8_t* q) {  // 16 bytes, the first one most significant
  return _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(q)), _mm_set_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
}
NVH_CLMUL_TARGET inline __m128i crc_fold(__m128i a, __m128i k, __m128i next) {  // hi64(a) * hi64(k) + lo64(a) * lo64(k) + next
  return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(a, k, 0x11), _mm_clmulepi64_si128(a, k, 0x00)), next);
}
NVH_CLMUL_TARGET uint32_t crc_clmul(uint32_t crc, const uint8_t* p, size_t n) {
  const __m128i k64 = _mm_set_epi64x(0x8833794cll, 0xe6228b11ll), k16 = _mm_set_epi64x(0xc5b9cd4cll, 0xe8a45605ll);
#define ld crc_ld
#define fold crc_fold
  __m128i a0 = _mm_xor_si128(ld(p), _mm_set_epi32((int)crc, 0, 0, 0)), a1 = ld(p + 16), a2 = ld(p + 32), a3 = ld(p + 48);
  p += 64;
  n -= 64;
  while (n >= 64) {
    a0 = fold(a0, k64, ld(p));
    a1 = fold(a1, k64, ld(p + 16));
    a2 = fold(a2, k64, ld(p + 32));
    a3 = fold(a3, k64, ld(p + 48));
    p += 64;
    n -= 64;
  }
  a1 = fold(a0, k16, a1);
  a2 = fold(a1, k16, a2);
  a3 = fold(a2, k16, a3);
  while (n >= 16) {
    a3 = fold(a3, k16, ld(p));
    p += 16;
    n -= 16;
  }
#undef ld
#undef fold
  uint8_t acc[16];
  _mm_storeu_si128(reinterpret_cast<__m128i*>(acc),
                   _mm_shuffle_epi8(a3, _mm_set_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)));
  return crc_table(crc_table(0, acc, 16), p, n);
}
#endif

uint32_t crc_update(uint32_t crc, const uint8_t* p, size_t n) {
#if defined(__x86_64__)
  static const bool clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("ssse3") && !std::getenv("NVH_NO_CLMUL");
  if (clmul && n >= 128) return crc_clmul(crc, p, n);
#endif
  return crc_table(crc, p, n);
}

// The page's checksum field (bytes 22..25) counts as zero (Ogg/PageReaderBase.cs:33-70): the header -- 27 bytes + the segment
// table, total >= that by the caller's bounds check -- goes through a copy with the field cleared, the body as it lies.
bool page_crc_ok(const uint8_t* pg, size_t total) {
  uint8_t hdr[27 + 255];
  const size_t hlen = 27 + (size_t)pg[26];
  std::memcpy(hdr, pg, hlen);
  hdr[22] = hdr[23] = hdr[24] = hdr[25] = 0;
  uint32_t crc = crc_update(0, hdr, hlen);
  crc = crc_update(crc, pg + hlen, total - hlen);
  uint32_t want = (uint32_t)pg[22] | ((uint32_t)pg[23] << 8) | ((uint32_t)pg[24] << 16) | ((uint32_t)pg[25] << 24);
  return crc == want;
}

}  // namespace


int ogg_demux(const uint8_t* bytes, size_t len, OggPackets& out, int stream_index, int* nstreams, bool want_pages, OggIndexMode* index_mode) {
  // Logical streams in the order their first page appears (Ogg/PageReader.cs:126-158): a page with a serial number that
  // has no reader opens a new stream (multiplexed streams interleave their pages, chained streams follow one another);
  // the end-of-stream page retires the serial, so a later page with the same number starts another stream; a page
  // without packets is refused and its serial ignored from then on (Ogg/PageReaderBase.cs:72-85).
  struct Logical {
    uint32_t serial = 0;
    std::vector<Page> pages;
    bool has_all_pages = false;
    // StreamPageReader.AddPage state (Ogg/StreamPageReader.cs:44-91)
    int32_t last_seq = 0;
    bool have_first_data_page = false;
    int64_t max_granule = 0;
    int first_data_page = -1;
    int error = NVH_OK;  // a granule rule of AddPage was broken: the stream is refused (and takes no further pages)
  };
  std::vector<Logical> streams;
  std::vector<std::pair<uint32_t, int>> active;  // serial -> index into streams
  std::vector<uint32_t> ignored;
  bool resync = false;
  size_t pos = 0;

  while (pos + 27 <= len) {
    const uint8_t* h = bytes + pos;
    if (!(h[0] == 0x4f && h[1] == 0x67 && h[2] == 0x67 && h[3] == 0x53)) {
      ++pos;
      resync = true;
      continue;
    }
    int seg_cnt = h[26];
    if (pos + 27 + (size_t)seg_cnt > len) {
      ++pos;
      resync = true;
      continue;
    }
    size_t data_len = 0;
    for (int s = 0; s < seg_cnt; s++) data_len += h[27 + s];
    size_t total = 27 + (size_t)seg_cnt + data_len;
    if (pos + total > len || (!index_mode && !page_crc_ok(h, total))) {
      ++pos;
      resync = true;
      continue;
    }
    uint32_t pg_serial = (uint32_t)h[14] | ((uint32_t)h[15] << 8) | ((uint32_t)h[16] << 16) | ((uint32_t)h[17] << 24);
    bool skip = false;
    for (uint32_t ig : ignored) skip = skip || ig == pg_serial;
    if (!skip) {
      Page pg;
      pg.data_off = pos + 27 + (size_t)seg_cnt;
      pg.flags = h[5];
      std::memcpy(&pg.granule, h + 6, 8);
      pg.resync = resync;
      int size = 0, off = 0;
      for (int s = 0; s < seg_cnt; s++) {
        int seg = h[27 + s];
        size += seg;
        if (seg < 255) {
          if (size > 0) {
            pg.pk_off.push_back(off);
            pg.pk_len.push_back(size);
            off += size;
          }
          size = 0;
        }
      }
      if (size > 0) {
        pg.continued = h[seg_cnt + 26] == 255;
        pg.pk_off.push_back(off);
        pg.pk_len.push_back(size);
      }
      int slot = -1;
      for (size_t a = 0; a < active.size(); a++)
        if (active[a].first == pg_serial) slot = (int)a;
      if (pg.pk_off.empty()) {
        // refused page: the serial is ignored from here on; a stream it would have opened does not come to exist
        ignored.push_back(pg_serial);
        if (slot >= 0) active.erase(active.begin() + slot);
      } else {
        if (slot < 0) {
          Logical lg;
          lg.serial = pg_serial;
          streams.push_back(std::move(lg));
          active.emplace_back(pg_serial, (int)streams.size() - 1);
          slot = (int)active.size() - 1;
        }
        Logical& lg = streams[(size_t)active[(size_t)slot].second];
        {
          // StreamPageReader.AddPage (Ogg/StreamPageReader.cs:50-86).  Granule sanity: the reference throws InvalidDataException
          // from inside its page reader; here the file is refused.  Resync mark: lost page sync, or a page sequence number
          // that does not follow the previous one ("out of order page / sequence jump, we're counting it as a resync").
          const int32_t seq = (int32_t)((uint32_t)h[18] | ((uint32_t)h[19] << 8) | ((uint32_t)h[20] << 16) | ((uint32_t)h[21] << 24));
          if (pg.granule != -1) {
            if (!lg.have_first_data_page && pg.granule > 0) {
              lg.have_first_data_page = true;
              lg.first_data_page = (int)lg.pages.size();
            }
            else if (lg.max_granule > pg.granule) lg.error = NVH_ERR_INVALID_DATA;  // "Granule Position regressed?!"
            lg.max_granule = pg.granule;
          } else if (lg.have_first_data_page && (!pg.continued || pg.pk_off.size() != 1)) {
            lg.error = NVH_ERR_INVALID_DATA;  // "Granule Position was -1 but page does not have exactly 1 continued packet."
          }
          if (lg.error != NVH_OK) {
            // the reference throws from inside its page reader when it gets here; only a caller that asks for THIS stream is told
            active.erase(active.begin() + slot);
            ignored.push_back(pg_serial);
            resync = false;
            pos += total;
            continue;
          }
          pg.resync = pg.resync || (lg.last_seq != 0 && (int32_t)((uint32_t)lg.last_seq + 1u) != seq);
          lg.last_seq = seq;
        }
        const bool eos_page = (pg.flags & 0x04) != 0;
        lg.pages.push_back(std::move(pg));
        if (eos_page) {
          lg.has_all_pages = true;
          active.erase(active.begin() + slot);
        }
      }
    }
    resync = false;
    pos += total;
  }
  if (nstreams) *nstreams = (int)streams.size();
  out.bytes.clear();
  out.offs.clear();
  out.granule.clear();
  out.flags.clear();
  if (stream_index < 0 || stream_index >= (int)streams.size()) {
    out.offs.push_back(0);
    return stream_index == 0 ? NVH_OK : NVH_ERR_ARGUMENT;  // an input without any page: an empty packet list, as before
  }
  if (streams[(size_t)stream_index].error != NVH_OK) return streams[(size_t)stream_index].error;
  const std::vector<Page>& pages = streams[(size_t)stream_index].pages;
  const bool has_all_pages = streams[(size_t)stream_index].has_all_pages;
  out.pages.clear();
  out.first_data_page = streams[(size_t)stream_index].first_data_page;
  out.has_all_pages = has_all_pages;
  out.max_granule = streams[(size_t)stream_index].max_granule;
  if (want_pages) {
    out.pages.resize(pages.size());
    for (size_t i = 0; i < pages.size(); i++) {
      const Page& pg = pages[i];
      OggPageInfo& pi = out.pages[i];
      pi.granule = pg.granule;
      pi.resync = pg.resync;
      pi.continuation = (pg.flags & 0x01) != 0;
      pi.continued = pg.continued;
      pi.packet_count = (int)pg.pk_off.size();
      pi.flat.assign(pg.pk_off.size(), -1);
      pi.len.resize(pg.pk_off.size());
      pi.head.assign(pg.pk_off.size() * 8, 0);
      for (size_t k = 0; k < pg.pk_off.size(); k++) {
        pi.len[k] = pg.pk_len[k];
        const int nb = pg.pk_len[k] < 8 ? pg.pk_len[k] : 8;
        std::memcpy(&pi.head[k * 8], bytes + pg.data_off + pg.pk_off[k], (size_t)nb);
      }
    }
  }

  out.bytes.clear();
  out.offs.clear();
  out.granule.clear();
  out.flags.clear();
  const int npages = (int)pages.size();
  int page_index = 0, packet_index = 0;
  int64_t payload = 0;
  // a fragment of the packet that is being assembled (index form: only up to head_bytes of it are kept, behind the headers)
  size_t mark = 0;
  auto append = [&](const uint8_t* p, size_t n) {
    payload += (int64_t)n;
    if (index_mode && (int)out.granule.size() >= index_mode->full_first) {
      const size_t have = out.bytes.size() - mark, cap = (size_t)index_mode->head_bytes;
      n = have >= cap ? 0 : (n < cap - have ? n : cap - have);
    }
    out.bytes.insert(out.bytes.end(), p, p + n);
  };
  while (page_index < npages) {
    const Page& pg = pages[(size_t)page_index];
    int64_t granule_pos = pg.granule;
    bool is_resync = pg.resync, is_continued = pg.continued;
    int packet_count = (int)pg.pk_off.size();
    bool is_last_packet;
    int final_page = page_index;
    mark = out.bytes.size();
    const int64_t payload_mark = payload;
    append(bytes + pg.data_off + pg.pk_off[(size_t)packet_index], (size_t)pg.pk_len[(size_t)packet_index]);
    if (is_continued && packet_index == packet_count - 1) {
      int cont = page_index;
      bool truncated = false;
      while (is_continued) {
        if (++cont >= npages) {
          truncated = true;
          break;
        }
        const Page& np = pages[(size_t)cont];
        granule_pos = np.granule;
        is_resync = np.resync;
        bool is_continuation = (np.flags & 0x01) != 0;
        is_continued = np.continued;
        packet_count = (int)np.pk_off.size();
        if (!is_continuation || is_resync) break;
        if (is_continued && packet_count > 1) is_continued = false;
        append(bytes + np.data_off + np.pk_off[0], (size_t)np.pk_len[0]);
      }
      if (truncated) {
        out.bytes.resize(mark);
        payload = payload_mark;
        break;
      }
      is_last_packet = packet_count == 1;
      final_page = cont;
    } else {
      is_last_packet = packet_index == packet_count - 1;
    }
    uint8_t fl = 0;
    int64_t gr = -1;
    if (is_resync) fl |= 2;
    if (is_last_packet) {
      gr = granule_pos < 0 ? -1 : granule_pos;
      if (has_all_pages && final_page == npages - 1) fl |= 1;
    }
    if (want_pages) out.pages[(size_t)page_index].flat[(size_t)packet_index] = (int32_t)out.granule.size();
    out.offs.push_back((int64_t)mark);
    out.granule.push_back(gr);
    out.flags.push_back(fl);
    if (final_page != page_index) {
      page_index = final_page;
      packet_index = 0;
    }
    if (packet_index == packet_count - 1) {
      ++page_index;
      packet_index = 0;
    } else {
      ++packet_index;
    }
  }
  out.offs.push_back((int64_t)out.bytes.size());
  if (index_mode) index_mode->payload_bytes = payload;
  return NVH_OK;
}

// ------------------------------------------------------------------------------------------------
// forward-only reader (non-seekable sources)
// ------------------------------------------------------------------------------------------------
// ForwardOnlyPageReader.AddPage + ForwardOnlyPacketProvider (Ogg/ForwardOnlyPageReader.cs:21-52,
// Ogg/ForwardOnlyPacketProvider.cs:36-67, 119-290): what VorbisReader gets from a stream that cannot seek.  Same page sync
// and CRC as above; the differences are the provider's:
//   * the beginning-of-stream page is a resync page; every other page is one when its sequence number is not the previous + 1
//     (no exemption for a previous number of 0), and there are no granule-position rules                       (:38-53)
//   * a page is refused only when all its lacing values are 0; zero-length packets inside a page ARE delivered  (:55-64, 270-284)
//   * a continuation page met at a packet start: resync, its partial first packet is skipped -- but only in the lacing table,
//     the data offset stays where it was, so the packets of that page are cut from the wrong bytes            (:147-165)
//   * the granule position (and end of stream) go to the last packet that is COMPLETE on its page; a packet that continues onto
//     following pages never gets either                                                                         (:176-231)
//   * a page that ends a continuation run abnormally (resync / not a continuation) is kept as the current page and its own
//     resync flag is forgotten                                                                                  (:196-228)
// The reference pulls pages on demand; with one logical stream per serial that is the same as having them all.
int ogg_demux_forward(const uint8_t* bytes, size_t len, OggPackets& out, int stream_index, int* nstreams) {
  struct FPage {
    const uint8_t* pg;  // page start (header)
    bool resync;
  };
  struct Logical {
    uint32_t serial = 0;
    std::vector<FPage> pages;
    int32_t last_seq = 0;
    bool ended = false;  // SetEndOfStream
  };
  std::vector<Logical> streams;
  std::vector<std::pair<uint32_t, int>> active;
  std::vector<uint32_t> ignored;
  bool resync = false;
  size_t pos = 0;
  while (pos + 27 <= len) {
    const uint8_t* h = bytes + pos;
    if (!(h[0] == 0x4f && h[1] == 0x67 && h[2] == 0x67 && h[3] == 0x53)) {
      ++pos;
      resync = true;
      continue;
    }
    const int seg_cnt = h[26];
    if (pos + 27 + (size_t)seg_cnt > len) {
      ++pos;
      resync = true;
      continue;
    }
    size_t data_len = 0;
    for (int s = 0; s < seg_cnt; s++) data_len += h[27 + s];
    const size_t total = 27 + (size_t)seg_cnt + data_len;
    if (pos + total > len || !page_crc_ok(h, total)) {
      ++pos;
      resync = true;
      continue;
    }
    const uint32_t pg_serial = (uint32_t)h[14] | ((uint32_t)h[15] << 8) | ((uint32_t)h[16] << 16) | ((uint32_t)h[17] << 24);
    bool skip = false;
    for (uint32_t ig : ignored) skip = skip || ig == pg_serial;
    if (!skip) {
      int slot = -1;
      for (size_t a = 0; a < active.size(); a++)
        if (active[a].first == pg_serial) slot = (int)a;
      const bool is_new = slot < 0;
      Logical fresh;
      fresh.serial = pg_serial;
      Logical& lg = is_new ? fresh : streams[(size_t)active[(size_t)slot].second];
      // ForwardOnlyPacketProvider.AddPage (:36-67)
      bool accept = true, pg_resync = resync;
      const int32_t seq = (int32_t)((uint32_t)h[18] | ((uint32_t)h[19] << 8) | ((uint32_t)h[20] << 16) | ((uint32_t)h[21] << 24));
      if (h[5] & 0x02) {  // BeginningOfStream
        if (lg.ended) accept = false;
        pg_resync = true;
        if (accept) lg.last_seq = seq;
      } else {
        pg_resync = pg_resync || seq != (int32_t)((uint32_t)lg.last_seq + 1u);
        lg.last_seq = seq;
      }
      if (accept && data_len == 0) accept = false;  // "there must be at least one packet with data"
      if (accept) {
        lg.pages.push_back(FPage{h, pg_resync});
        if (is_new) {
          streams.push_back(std::move(fresh));
          active.emplace_back(pg_serial, (int)streams.size() - 1);
          slot = (int)active.size() - 1;
        }
        if (h[5] & 0x04) {  // EndOfStream: SetEndOfStream, the reader forgets the provider (ForwardOnlyPageReader.cs:28-33)
          streams[(size_t)active[(size_t)slot].second].ended = true;
          active.erase(active.begin() + slot);
        }
      } else {
        ignored.push_back(pg_serial);  // PageReaderBase.AddPage (:72-85)
        if (!is_new) active.erase(active.begin() + slot);
      }
    }
    resync = false;
    pos += total;
  }
  if (nstreams) *nstreams = (int)streams.size();
  out.bytes.clear();
  out.offs.clear();
  out.granule.clear();
  out.flags.clear();
  out.pages.clear();
  if (stream_index < 0 || stream_index >= (int)streams.size()) {
    out.offs.push_back(0);
    return stream_index == 0 ? NVH_OK : NVH_ERR_ARGUMENT;
  }
  const Logical& lg = streams[(size_t)stream_index];
  const std::vector<FPage>& pages = lg.pages;

  // ---- ForwardOnlyPacketProvider.GetPacket (:119-246), called until it returns false ----
  size_t next_page = 0;            // the queue
  const uint8_t* page_buf = nullptr;  // _pageBuf
  int st_packet_index = 0x7fffffff, st_data_start = 0;
  auto packet_length = [](const uint8_t* pb, int& packet_index) {  // GetPacketLength (:270-284)
    int l = 0;
    while (packet_index < pb[26] + 27 && pb[packet_index] == 255) {
      l += pb[packet_index];
      ++packet_index;
    }
    if (packet_index < pb[26] + 27) {
      l += pb[packet_index];
      ++packet_index;
    }
    return l;
  };
  auto read_next_page = [&](const uint8_t*& pb, bool& is_resync, int& data_start, int& packet_index, bool& is_cont, bool& is_cntd) {
    if (next_page >= pages.size()) return false;  // the queue is empty and the reader has nothing more
    pb = pages[next_page].pg;
    is_resync = pages[next_page].resync;
    ++next_page;
    data_start = pb[26] + 27;
    packet_index = 27;
    is_cont = (pb[5] & 0x01) != 0;
    is_cntd = pb[26 + pb[26]] == 255;
    return true;
  };
  for (;;) {
    const uint8_t* pb;
    bool is_resync, is_cont, is_cntd;
    int data_start, packet_index;
    if (page_buf != nullptr && st_packet_index < 27 + page_buf[26]) {
      pb = page_buf;
      is_resync = false;
      data_start = st_data_start;
      packet_index = st_packet_index;
      is_cont = false;
      is_cntd = pb[26 + pb[26]] == 255;
    } else if (!read_next_page(pb, is_resync, data_start, packet_index, is_cont, is_cntd)) {
      break;
    }
    const bool is_first = packet_index == 27;
    if (is_cont && is_first) {
      is_resync = true;
      (void)packet_length(pb, packet_index);  // "skip the first packet; it's a partial" -- the data offset is not moved
      if (packet_index == 27 + pb[26]) continue;  // "we'll just recurse and try again": _pageBuf is still the exhausted old page
    }
    const int data_len = packet_length(pb, packet_index);
    const size_t mark = out.bytes.size();
    {
      // the slice is taken from the page array as the lacing says, even where the stale offset runs it past the page's end
      // in the file image (then whatever follows is read; the managed array would fault: kept inside the input here)
      const uint8_t* src = pb + data_start;
      size_t avail = (size_t)((bytes + len) - src);
      size_t take = (size_t)data_len <= avail ? (size_t)data_len : avail;
      out.bytes.insert(out.bytes.end(), src, src + take);
      out.bytes.insert(out.bytes.end(), (size_t)data_len - take, (uint8_t)0);
    }
    data_start += data_len;
    bool is_last = packet_index == 27 + pb[26];
    if (is_cntd) {
      if (is_last) {
        is_last = false;
      } else {
        int pi = packet_index;
        (void)packet_length(pb, pi);
        is_last = pi == 27 + pb[26];
      }
    }
    bool is_eos = false;
    int64_t gr = -1;
    bool has_gr = false;
    if (is_last) {
      std::memcpy(&gr, pb + 6, 8);
      has_gr = true;
      // (_isEndOfStream && _pageQueue.Count == 0 adds nothing when pages are pulled one at a time)
      if (pb[5] & 0x04) is_eos = true;
    } else {
      while (is_cntd && packet_index == 27 + pb[26]) {
        const uint8_t* nb;
        bool n_resync, n_cont, n_cntd;
        int n_start, n_index;
        const bool got = read_next_page(nb, n_resync, n_start, n_index, n_cont, n_cntd);
        if (got) {  // the out parameters are written whether or not the page continues the packet
          pb = nb;
          is_resync = n_resync;
          data_start = n_start;
          packet_index = n_index;
          is_cont = n_cont;
          is_cntd = n_cntd;
        } else {
          pb = nullptr;  // ReadNextPage sets pageBuf = null on failure
          is_resync = false;
          data_start = 0;
          packet_index = 0;
          is_cont = false;
          is_cntd = false;
        }
        if (got && !is_resync && is_cont) {
          const int cont_sz = packet_length(pb, packet_index);
          out.bytes.insert(out.bytes.end(), pb + data_start, pb + data_start + cont_sz);
          data_start += cont_sz;
        } else {
          break;
        }
      }
    }
    uint8_t fl = 0;
    if (is_resync) fl |= 2;
    if (is_eos) fl |= 1;
    out.offs.push_back((int64_t)mark);
    out.granule.push_back(has_gr ? gr : -1);
    out.flags.push_back(fl);
    page_buf = pb;
    st_data_start = data_start;
    st_packet_index = packet_index;
    if (page_buf == nullptr) {
      // _pageBuf = null: the next call goes straight to ReadNextPage, which has nothing left
      st_packet_index = 0x7fffffff;
    }
  }
  out.offs.push_back((int64_t)out.bytes.size());
  return NVH_OK;
}

// ------------------------------------------------------------------------------------------------
// seek search
// ------------------------------------------------------------------------------------------------
namespace {

struct SeekCtx {
  const OggPackets& ix;
  OggGranuleCount count;
  void* user;
  int npages() const { return (int)ix.pages.size(); }
};

// What PacketProvider.CreatePacket(.., advance: false, ..) (Ogg/PacketProvider.cs:324-400) leaves of a packet as far as
// GetPacketGranules is concerned: its first bytes (the slot's fragment, then the slot-0 fragments of the pages the
// continuation walk adds) and the IsResync flag the walk ends with.  false = the method returns null.
bool create_packet(const SeekCtx& c, int page_index, int packet_index, bool is_resync, bool is_continued, int packet_count,
                   uint8_t head[8], int* head_len, bool* resync_out, int* fault) {
  const OggPageInfo& pg = c.ix.pages[(size_t)page_index];
  if (packet_index < 0 || packet_index >= pg.packet_count) {  // GetPagePackets(pageIndex)[packetIndex]
    *fault = NVH_ERR_RUNTIME;
    return false;
  }
  int n = pg.len[(size_t)packet_index] < 8 ? pg.len[(size_t)packet_index] : 8;
  std::memcpy(head, &pg.head[(size_t)packet_index * 8], (size_t)n);
  if (is_continued && packet_index == packet_count - 1) {
    int cont = page_index;
    while (is_continued) {
      if (++cont >= c.npages()) return false;  // "no more pages?  In any case, we can't satisfy the request"
      const OggPageInfo& np = c.ix.pages[(size_t)cont];
      is_resync = np.resync;
      is_continued = np.continued;
      packet_count = np.packet_count;
      if (!np.continuation || is_resync) break;
      if (is_continued && packet_count > 1) is_continued = false;
      const int more = np.len[0] < 8 - n ? np.len[0] : 8 - n;
      if (more > 0) {
        std::memcpy(head + n, &np.head[0], (size_t)more);
        n += more;
      }
    }
  }
  *head_len = n;
  *resync_out = is_resync;
  return true;
}

// Ogg/PacketProvider.cs:224-260
bool is_vorbis_bug_diff(int64_t diff) {
  if (diff < 0) diff = -diff;
  int64_t temp = diff;
  int short_bits = 0;
  while (temp > 0 && (temp & 1) == 0) {
    ++short_bits;
    temp >>= 1;
  }
  int long_bits = short_bits;
  while ((temp & 1) == 1) {
    ++long_bits;
    temp >>= 1;
  }
  // (1 << longBlockBits) - (1 << shortBlockBits) is 32-bit arithmetic in the reference: shift counts taken modulo 32, the
  // difference wrapped to int, then widened for the comparison
  const uint32_t hi = 1u << (long_bits & 31), lo = 1u << (short_bits & 31);
  return temp == 0 && diff == (int64_t)(int32_t)(hi - lo);
}

// StreamPageReader.FindPage with every page already read (Ogg/StreamPageReader.cs:122-264); -1 = ArgumentOutOfRangeException
int find_page(const SeekCtx& c, int64_t granule_pos) {
  const int n = c.npages();
  if (granule_pos == 0) return c.ix.first_data_page;
  int page_index = -1;
  const int last = n - 1;
  if (last < 0) return -1;
  const int64_t last_gp = c.ix.pages[(size_t)last].granule;
  if (granule_pos < last_gp) {
    // FindPageBisection(granulePos, FindFirstDataPage(), lastPageIndex, pageGP) (:232-264)
    int low = c.ix.first_data_page, high = last;
    int64_t low_gp = 0, high_gp = last_gp;
    int dist;
    while ((dist = high - low) > 0) {
      const int index = low + (int)((double)dist * ((double)(granule_pos - low_gp) / (double)(high_gp - low_gp)));
      if (index < 0 || index >= n) return -2;  // _pageOffsets[index] faults (a stream without a data page: low == -1)
      const int64_t idx_gp = c.ix.pages[(size_t)index].granule;
      if (idx_gp > granule_pos) {
        high = index;
        high_gp = idx_gp;
      } else if (idx_gp < granule_pos) {
        low = index + 1;
        low_gp = idx_gp + 1;
      } else {
        return index + 1;
      }
    }
    page_index = low;
  } else if (granule_pos > last_gp) {
    // FindPageForward (:171-199): no page is left to read, so the walk ends at once; the reader then knows it has all
    // pages (GetNextPageGranulePos, :201-230) and MaxGranulePosition < granulePos decides
    page_index = last + 1;
    if (c.ix.max_granule < granule_pos) page_index = -1;
  } else {
    page_index = last + 1;
  }
  return page_index;
}

}  // namespace

int ogg_seek(const OggPackets& ix, OggGranuleCount granule_count, void* user, int64_t granule_pos, int pre_roll, int64_t* packet,
             int64_t* granule_out) {
  const SeekCtx c{ix, granule_count, user};
  const int n = c.npages();
  int fault = NVH_OK;
  int page_index = find_page(c, granule_pos);
  if (page_index == -2) return NVH_ERR_RUNTIME;
  if (page_index == -1) return NVH_ERR_ARGUMENT;  // ArgumentOutOfRangeException(nameof(granulePos)) (:157-160)

  // ---- FindPacket(pageIndex, preRoll, ref granulePos, ..) (Ogg/PacketProvider.cs:204-222) ----
  // GetPreviousPageInfo (:74-108)
  int64_t last_page_gp = 0;
  int last_page_packet_length = 0, first_real_packet = 0;
  if (page_index > 0) {
    if (page_index - 1 >= n) return NVH_ERR_INVALID_DATA;  // "Could not get preceding page?!"
    const OggPageInfo& prev = ix.pages[(size_t)(page_index - 1)];
    last_page_gp = prev.granule;
    if (page_index > ix.first_data_page) {
      uint8_t head[8];
      int hl = 0;
      bool rs = false;
      // the last slot of the previous page: "either a continued packet OR the last packet of the last page"
      if (!create_packet(c, page_index - 1, prev.packet_count - 1, false, prev.continued, prev.packet_count, head, &hl, &rs, &fault))
        return fault != NVH_OK ? fault : NVH_ERR_INVALID_DATA;  // "Could not find end of continuation!"
      last_page_packet_length = granule_count(user, head, hl, rs);
    }
    first_real_packet = prev.continued ? 1 : 0;
  }
  // GetTargetPageInfo (:110-146)
  if (page_index < 0 || page_index >= n) return NVH_ERR_INVALID_DATA;  // "Could not get found page?!"
  const OggPageInfo& pg = ix.pages[(size_t)page_index];
  int packet_count = pg.packet_count;
  if (pg.continued) packet_count--;  // "if continued, the last packet index doesn't apply"
  std::vector<int64_t> gps((size_t)(packet_count > 0 ? packet_count : 0));
  int64_t end_gp = pg.granule;
  for (int i = packet_count - 1; i >= first_real_packet; i--) {
    gps[(size_t)i] = end_gp;
    uint8_t head[8];
    int hl = 0;
    bool rs = false;
    // (the reduced packet count and the page's continued flag go in as they are: the last complete packet of a continued page
    // is walked as if it were the continued one, and takes its resync flag from the next page)
    if (!create_packet(c, page_index, i, i == 0 && pg.resync, pg.continued, packet_count, head, &hl, &rs, &fault))
      return fault != NVH_OK ? fault : NVH_ERR_INVALID_DATA;  // "Could not find end of continuation!"
    end_gp -= granule_count(user, head, hl, rs);
  }
  if (first_real_packet == 1) {
    if (gps.empty()) return NVH_ERR_RUNTIME;  // gps[0] of an empty array
    gps[0] = end_gp;
    end_gp -= last_page_packet_length;
  }
  // FindPacket(pageIndex, gps, endGP, lastPageGranulePos, lastPagePacketLength, ref granulePos) (:148-202)
  int packet_index = -2;
  if (end_gp != last_page_gp) {
    const int64_t diff = end_gp - last_page_gp;
    if (is_vorbis_bug_diff(diff)) {
      if (diff > 0) {
        // the last packet of the previous page is a long block libvorbis mis-counted
        if (granule_pos <= end_gp) {
          granule_pos = end_gp - last_page_packet_length;
          packet_index = -1;
        }
      } else {
        for (int64_t& g : gps) g -= diff;  // "our pageGranulePos is wrong, so adjust everything"
      }
    } else if (page_index > ix.first_data_page) {
      return NVH_ERR_INVALID_DATA;  // "GranulePos mismatch"
    }
  }
  if (packet_index == -2) {
    for (size_t i = 0; i < gps.size(); i++) {
      if (gps[i] >= granule_pos) {
        granule_pos = i == 0 ? end_gp : gps[i - 1];
        packet_index = (int)i;
        break;
      }
    }
    if (packet_index == -2) return NVH_ERR_INVALID_DATA;  // "Could not find seek packet?!"
  }
  // the pre-roll, "but only if we're not seeking into the first packet, which is its own preRoll" (:216-220)
  if (end_gp > 0 || packet_index > 1) packet_index -= pre_roll;

  // ---- NormalizePacketIndex (:262-295): false = ArgumentOutOfRangeException (:63-66) ----
  {
    bool is_resync = pg.resync, is_continuation = pg.continuation;
    int pg_idx = page_index, pkt_idx = packet_index;
    while (pkt_idx < (is_continuation ? 1 : 0)) {
      if (is_continuation && is_resync) return NVH_ERR_ARGUMENT;  // can't merge across resync
      const bool was_continuation = is_continuation;
      if (--pg_idx < 0) return NVH_ERR_ARGUMENT;
      const OggPageInfo& pp = ix.pages[(size_t)pg_idx];
      is_resync = pp.resync;
      is_continuation = pp.continuation;
      if (was_continuation && !pp.continued) return NVH_ERR_ARGUMENT;  // continuation flags do not match
      pkt_idx += pp.packet_count - (was_continuation ? 1 : 0);
    }
    page_index = pg_idx;
    packet_index = pkt_idx;
  }
  // (page, packet) -> position in the demuxed list; a slot no packet starts at makes GetNextPacket fault or stitch garbage
  const OggPageInfo& fin = ix.pages[(size_t)page_index];
  if (packet_index >= fin.packet_count || fin.flat[(size_t)packet_index] < 0) return NVH_ERR_RUNTIME;
  *packet = fin.flat[(size_t)packet_index];
  *granule_out = granule_pos;
  return NVH_OK;
}

}  // namespace nvh
