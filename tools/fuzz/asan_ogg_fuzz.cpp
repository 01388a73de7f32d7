// asan_ogg_fuzz.cpp -- AddressSanitizer / UBSan run of the container code (host_ogg.cpp: both readers, page table, seek search) on
// mutated files.  No GPU, no HIP:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I nvorbis_amd/csrc -I include \
//       tools/fuzz/asan_ogg_fuzz.cpp nvorbis_amd/csrc/host_ogg.cpp -o /tmp/asan_ogg_fuzz && /tmp/asan_ogg_fuzz tests/golden/*.ogg
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <vector>

#include "host_ogg.h"
#include "nvorbis_hip.h"

static int fake_granules(void*, const uint8_t* head, int len, bool is_resync) {
  if (is_resync || len <= 0 || (head[0] & 1)) return 0;
  return (head[0] & 2) ? 1024 - 448 * ((head[0] >> 2) & 1) : 128;
}

static uint32_t crc_of(const std::vector<uint8_t>& pg) {
  uint32_t crc = 0;
  for (size_t i = 0; i < pg.size(); i++) {
    uint8_t b = (i >= 22 && i < 26) ? 0 : pg[i];
    crc ^= (uint32_t)b << 24;
    for (int k = 0; k < 8; k++) crc = (crc & 0x80000000u) ? (crc << 1) ^ 0x04c11db7u : (crc << 1);
  }
  return crc;
}

int main(int argc, char** argv) {
  long cases = 0, seeks = 0;
  for (int a = 1; a < argc; a++) {
    std::ifstream f(argv[a], std::ios::binary);
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (data.size() > 400000) data.resize(400000);
    for (int trial = 0; trial < 400; trial++) {
      std::mt19937 rng((unsigned)(trial * 7919 + a));
      std::vector<uint8_t> bad = data;
      // page-level mutations with the checksum repaired, then raw damage
      size_t pos = 0;
      int page = 0;
      while (pos + 27 <= bad.size() && std::memcmp(&bad[pos], "OggS", 4) == 0) {
        int nseg = bad[pos + 26];
        if (pos + 27 + (size_t)nseg > bad.size()) break;
        size_t len = 27 + (size_t)nseg;
        for (int s = 0; s < nseg; s++) len += bad[pos + 27 + (size_t)s];
        if (pos + len > bad.size()) break;
        if (page >= 2 && rng() % 8 == 0) {
          switch (rng() % 5) {
            case 0: bad[pos + 5] ^= (uint8_t)(1u << (rng() % 3)); break;
            case 1: { int64_t g = (int64_t)(rng() % 3 == 0 ? -1 : (int64_t)(rng() % 600000)); std::memcpy(&bad[pos + 6], &g, 8); break; }
            case 2: bad[pos + 18] = (uint8_t)(bad[pos + 18] + rng() % 4); break;
            case 3: if (nseg > 1) { size_t k = pos + 27 + rng() % (unsigned)(nseg - 1); if (bad[k] + bad[k + 1] <= 255) { /* merge would need a shorter table: turn a terminator into a continuation instead */ } bad[k] = 255; } break;
            default: break;
          }
          std::vector<uint8_t> pg(bad.begin() + (long)pos, bad.begin() + (long)(pos + len));
          // the lacing table may now claim more data than the page has: recompute what the reader will see and fix the CRC over it
          size_t claim = 27 + (size_t)nseg;
          for (int s = 0; s < nseg; s++) claim += pg[27 + (size_t)s];
          if (claim == len) {
            uint32_t c = crc_of(pg);
            std::memcpy(&bad[pos + 22], &c, 4);
          }
        }
        pos += len;
        page++;
      }
      int raw = (int)(rng() % 4);
      for (int k = 0; k < raw; k++) bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8));
      if (rng() % 10 == 0) bad.resize(bad.size() / 2 + rng() % (bad.size() / 2));
      for (int stream = 0; stream < 2; stream++) {
        nvh::OggPackets pk, fw;
        int ns = 0;
        int rc = nvh::ogg_demux(bad.data(), bad.size(), pk, stream, &ns, true);
        (void)nvh::ogg_demux_forward(bad.data(), bad.size(), fw, stream, &ns);
        cases++;
        if (rc != NVH_OK || pk.pages.empty()) continue;
        for (int k = 0; k < 40; k++) {
          int64_t g = k < 8 ? (int64_t)k * 97 : (int64_t)(rng() % (uint64_t)(pk.max_granule + 3 > 0 ? pk.max_granule + 3 : 3));
          int64_t p = 0, go = 0;
          (void)nvh::ogg_seek(pk, fake_granules, nullptr, g, (int)(rng() % 2), &p, &go);
          seeks++;
        }
      }
    }
  }
  std::printf("asan_ogg_fuzz: %ld demux cases, %ld seeks, no sanitizer report\n", cases, seeks);
  return 0;
}
