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

int ogg_demux(const uint8_t* bytes, size_t len, OggPackets& out);

}  // namespace nvh
