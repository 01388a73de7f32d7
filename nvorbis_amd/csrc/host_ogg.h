// host_ogg.h -- packet list produced by the minimal forward-only Ogg demux (host_ogg.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace nvh {

struct OggPackets {
  std::vector<uint8_t> bytes;
  std::vector<int64_t> offs;      // n + 1 byte offsets
  std::vector<int64_t> granule;   // -1 = packet carries no granule position
  std::vector<uint8_t> flags;     // NVH_PKT_EOS | NVH_PKT_RESYNC
};

// Packets of logical stream `stream_index` (0 = the first one whose page appears); *nstreams = how many there are.
int ogg_demux(const uint8_t* bytes, size_t len, OggPackets& out, int stream_index = 0, int* nstreams = nullptr);

}  // namespace nvh
