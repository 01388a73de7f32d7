// host_parse_bench.cpp -- where the host parser's thread spends its time per frame: the bit parse (host_parse.cpp) and the slab
// writer (host_slab.cpp), no GPU involved.  Packets of an .ogg file, the long ones, cycled.
//   g++ -O2 -std=c++17 -Invorbis_amd/csrc -Iinclude tools/host_parse_bench.cpp nvorbis_amd/csrc/host_*.cpp -o /tmp/host_parse_bench
//   /tmp/host_parse_bench tests/golden/3test.ogg [frames per batch] [batches]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "host_ogg.h"
#include "host_parse.h"
#include "host_setup.h"
#include "host_slab.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> bytes;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
  fclose(f);
  const int per = argc > 2 ? atoi(argv[2]) : 4096, batches = argc > 3 ? atoi(argv[3]) : 8;
  nvh::OggPackets P;
  nvh::ogg_demux(bytes.data(), bytes.size(), P);
  nvh::Setup S;
  auto pkt = [&](size_t i) { return P.bytes.data() + P.offs[i]; };
  auto len = [&](size_t i) { return (int)(P.offs[i + 1] - P.offs[i]); };
  if (S.parse_id(pkt(0), len(0)) || S.parse_comment_sig(pkt(1), len(1)) || S.parse_setup(pkt(2), len(2))) return 1;
  nvh::SlabSetup X;
  std::vector<float> vq;
  std::vector<uint32_t> lattice;
  nvh::build_book_directory(S, X, vq, lattice);
  X.lattice = lattice;
  nvh::classify_residues(S, X, false);
  std::vector<size_t> longs;
  const size_t npk = P.offs.size() - 1;
  for (size_t i = 3; i < npk; i++)
    if (len(i) > 250) longs.push_back(i);
  if (longs.empty()) return 1;
  nvh::StreamParser sp(&S);
  nvh::FrameBatch B;
  nvh::SlabBatch SB;
  double t_parse = 0, t_slab = 0;
  size_t frames = 0, slab_bytes = 0, k = 0;
  for (int b = 0; b < batches; b++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < per; i++, k++) sp.push_packet(pkt(longs[k % longs.size()]), len(longs[k % longs.size()]), -1, 0, B);
    auto t1 = std::chrono::steady_clock::now();
    const int rc = nvh::build_slabs(S, X, B, SB);
    auto t2 = std::chrono::steady_clock::now();
    if (rc) { printf("build_slabs -> %d\n", rc); return 1; }
    if (b > 0) {  // (the first batch grows the vectors)
      t_parse += std::chrono::duration<double, std::micro>(t1 - t0).count();
      t_slab += std::chrono::duration<double, std::micro>(t2 - t1).count();
      frames += B.frames.size();
      slab_bytes += SB.data.size() * 16;
    }
    B.clear();
    sp.begin_batch();
  }
  printf("%zu frames: bit parse %.2f us per frame, slab writer %.2f us per frame (%.0f bytes per slab), digits %s\n", frames, t_parse / frames,
         t_slab / frames, (double)slab_bytes / frames, X.digits_ok ? "on" : "off");
  return 0;
}
