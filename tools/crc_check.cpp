// crc_check.cpp -- the carry-less-multiplication form of the Ogg page checksum (host_ogg.cpp: crc_clmul) against the table form on random
// buffers, lengths around the block sizes and initial values, then both rates.  g++ -O2 -std=c++17 -I nvorbis_amd/csrc tools/crc_check.cpp
// (tests/test_host_logic.py::test_clmul_page_checksum_equals_the_table builds and runs it)
#include "host_ogg.cpp"
#include <cstdio>
#include <random>
#include <chrono>
using namespace nvh;
int main() {
  std::mt19937 rng(7);
  std::vector<uint8_t> buf(70000);
  for (auto& b : buf) b = (uint8_t)rng();
  int bad = 0, n = 0;
  for (size_t len : {128u, 129u, 143u, 144u, 191u, 192u, 200u, 255u, 256u, 1000u, 4096u, 4400u, 65307u}) {
    for (int rep = 0; rep < 20; rep++) {
      const size_t off = rng() % 64;
      const uint32_t init = rep ? (uint32_t)rng() : 0u;
      const uint32_t a = crc_table(init, buf.data() + off, len), b = crc_clmul(init, buf.data() + off, len);
      bad += a != b; ++n;
    }
  }
  printf("%d of %d differ\n", bad, n);
  for (int w = 0; w < 2; w++) {
    auto t0 = std::chrono::steady_clock::now();
    uint32_t acc = 0;
    for (int r = 0; r < 20000; r++) acc ^= w ? crc_clmul(r, buf.data(), 65307) : crc_table(r, buf.data(), 65307);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %.2f GB/s (%u)\n", w ? "clmul" : "table", 20000 * 65307.0 / dt / 1e9, acc);
  }
  return bad != 0;
}
