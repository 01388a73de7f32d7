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
// The page-level seek search and the libvorbis granule workaround are not implemented.
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

struct CrcTable {
  uint32_t t[256];
  CrcTable() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t s = i << 24;
      for (int j = 0; j < 8; ++j) s = (s << 1) ^ (s >= (1u << 31) ? 0x04c11db7u : 0u);
      t[i] = s;
    }
  }
};

bool page_crc_ok(const uint8_t* pg, size_t total) {
  static const CrcTable tab;
  uint32_t crc = 0;
  for (size_t i = 0; i < total; i++) {
    uint8_t b = (i >= 22 && i < 26) ? 0 : pg[i];
    crc = (crc << 8) ^ tab.t[b ^ (crc >> 24)];
  }
  uint32_t want = (uint32_t)pg[22] | ((uint32_t)pg[23] << 8) | ((uint32_t)pg[24] << 16) | ((uint32_t)pg[25] << 24);
  return crc == want;
}

}  // namespace


int ogg_demux(const uint8_t* bytes, size_t len, OggPackets& out, int stream_index, int* nstreams) {
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
    if (pos + total > len || !page_crc_ok(h, total)) {
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
            if (!lg.have_first_data_page && pg.granule > 0) lg.have_first_data_page = true;
            else if (lg.max_granule > pg.granule) return NVH_ERR_INVALID_DATA;  // "Granule Position regressed?!"
            lg.max_granule = pg.granule;
          } else if (lg.have_first_data_page && (!pg.continued || pg.pk_off.size() != 1)) {
            return NVH_ERR_INVALID_DATA;  // "Granule Position was -1 but page does not have exactly 1 continued packet."
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
  const std::vector<Page>& pages = streams[(size_t)stream_index].pages;
  const bool has_all_pages = streams[(size_t)stream_index].has_all_pages;

  out.bytes.clear();
  out.offs.clear();
  out.granule.clear();
  out.flags.clear();
  const int npages = (int)pages.size();
  int page_index = 0, packet_index = 0;
  while (page_index < npages) {
    const Page& pg = pages[(size_t)page_index];
    int64_t granule_pos = pg.granule;
    bool is_resync = pg.resync, is_continued = pg.continued;
    int packet_count = (int)pg.pk_off.size();
    bool is_last_packet;
    int final_page = page_index;
    size_t mark = out.bytes.size();
    out.bytes.insert(out.bytes.end(), bytes + pg.data_off + pg.pk_off[(size_t)packet_index],
                     bytes + pg.data_off + pg.pk_off[(size_t)packet_index] + pg.pk_len[(size_t)packet_index]);
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
        out.bytes.insert(out.bytes.end(), bytes + np.data_off + np.pk_off[0], bytes + np.data_off + np.pk_off[0] + np.pk_len[0]);
      }
      if (truncated) {
        out.bytes.resize(mark);
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
  return NVH_OK;
}

}  // namespace nvh
