// asan_parse_fuzz.cpp -- AddressSanitizer / UBSan run of the bit-consuming host code (host_setup.cpp, host_parse.cpp) on mutated
// header and audio packets of the shipped files.  No GPU, no HIP:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I nvorbis_amd/csrc -I include \
//       tools/fuzz/asan_parse_fuzz.cpp nvorbis_amd/csrc/host_ogg.cpp nvorbis_amd/csrc/host_setup.cpp nvorbis_amd/csrc/host_parse.cpp \
//       -o /tmp/asan_parse_fuzz && /tmp/asan_parse_fuzz tests/golden/*.ogg
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <memory>
#include <random>
#include <vector>

#include "host_ogg.h"
#include "host_parse.h"
#include "host_setup.h"
#include "nvorbis_hip.h"

int main(int argc, char** argv) {
  long packets = 0, setups = 0, setups_ok = 0, errors = 0;
  for (int a = 1; a < argc; a++) {
    std::ifstream f(argv[a], std::ios::binary);
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    nvh::OggPackets pk;
    if (nvh::ogg_demux(data.data(), data.size(), pk, 0, nullptr) != NVH_OK || pk.granule.size() < 4) continue;
    auto pkt = [&](size_t i) { return std::vector<uint8_t>(pk.bytes.begin() + pk.offs[i], pk.bytes.begin() + pk.offs[i + 1]); };
    const size_t n = pk.granule.size();
    for (int trial = 0; trial < 60; trial++) {
      std::mt19937 rng((unsigned)(trial * 104729 + a));
      std::vector<uint8_t> id = pkt(0), cm = pkt(1), st = pkt(2);
      if (trial % 3 == 1)  // damaged setup header: a few bit flips anywhere behind the signature
        for (int k = 0; k < 1 + (int)(rng() % 3); k++) st[7 + rng() % (st.size() - 7)] ^= (uint8_t)(1u << (rng() % 8));
      if (trial % 3 == 2 && st.size() > 64) st.resize(64 + rng() % (st.size() - 64));
      auto S = std::make_unique<nvh::Setup>();
      setups++;
      if (S->parse_id(id.data(), (int)id.size()) != NVH_OK) continue;
      if (S->parse_comment_sig(cm.data(), (int)cm.size()) != NVH_OK) continue;
      if (S->parse_setup(st.data(), (int)st.size()) != NVH_OK) continue;
      setups_ok++;
      for (int light = 0; light < 2; light++) {
        nvh::StreamParser sp(S.get());
        sp.set_light(light != 0);
        nvh::FrameBatch fb;
        const size_t last = n < 260 ? n : 260;
        for (size_t i = 3; i < last; i++) {
          std::vector<uint8_t> p = pkt(i);
          const unsigned r = rng() % 10;
          if (r == 0 && !p.empty()) p.resize(rng() % p.size());
          else if (r == 1 && !p.empty()) p[rng() % p.size()] ^= (uint8_t)(1u << (rng() % 8));
          else if (r == 2) for (auto& b : p) b = (uint8_t)rng();
          int rc = sp.push_packet(p.data(), (int)p.size(), pk.granule[i], pk.flags[i], fb);
          if (rc != NVH_OK) errors++;
          packets++;
          if (fb.frames.size() > 64) {
            fb.clear();
            sp.begin_batch();
          }
        }
        (void)sp.push_end(fb);
      }
    }
  }
  std::printf("asan_parse_fuzz: %ld setups (%ld accepted), %ld packets (%ld refused the way the reference throws), no sanitizer report\n",
              setups, setups_ok, packets, errors);
  return 0;
}
