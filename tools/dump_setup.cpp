// dump_setup.cpp -- developer aid: print the codebook / residue geometry of an Ogg Vorbis file and how the
// residue work of its packets spreads over books (drives the specialisation choices of the spectrum kernel).
//   g++ -O2 -std=c++17 -Invorbis_amd/csrc -Iinclude tools/dump_setup.cpp nvorbis_amd/csrc/host_*.cpp -o /tmp/dump_setup
#include <cstdio>
#include <map>
#include <vector>

#include "host_ogg.h"
#include "host_parse.h"
#include "host_setup.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> bytes;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
  fclose(f);
  nvh::OggPackets P;
  nvh::ogg_demux(bytes.data(), bytes.size(), P);
  nvh::Setup S;
  auto pkt = [&](size_t i) { return P.bytes.data() + P.offs[i]; };
  auto len = [&](size_t i) { return (int)(P.offs[i + 1] - P.offs[i]); };
  if (S.parse_id(pkt(0), len(0)) || S.parse_comment_sig(pkt(1), len(1)) || S.parse_setup(pkt(2), len(2))) return 1;
  printf("channels %d blocks %d/%d books %zu floors %zu residues %zu mappings %zu modes %zu\n", S.channels, S.block0, S.block1,
         S.books.size(), S.floors.size(), S.residues.size(), S.mappings.size(), S.modes.size());
  for (size_t i = 0; i < S.residues.size(); i++) {
    const nvh::Residue& r = S.residues[i];
    printf("residue %zu: type %d begin %d end %d psize %d classes %d classbook %d (dim %d)\n", i, r.type, r.begin, r.end,
           r.partition_size, r.classifications, r.class_book, S.books[r.class_book].dimensions);
    for (int c = 0; c < r.classifications; c++) {
      printf("  class %d:", c);
      for (int k = 0; k < NVH_MAX_STAGES; k++)
        if (r.books[c][k] >= 0) {
          const nvh::Codebook& b = S.books[r.books[c][k]];
          printf("  s%d=book%d(dim %d, entries %d, map %d, lattice %d)", k, r.books[c][k], b.dimensions, b.entries, b.map_type,
                 b.lattice_values);
        }
      printf("\n");
    }
  }
  nvh::StreamParser sp(&S);
  nvh::FrameBatch B;
  for (size_t i = 3; i + 1 < P.offs.size(); i++) sp.push_packet(pkt(i), len(i), P.granule[i], P.flags[i], B);
  std::map<int, long> ops_by_book, ent_by_book;
  for (size_t o = 0; o < B.ops.size(); o++) {
    ops_by_book[B.ops[o].book]++;
  }
  printf("frames %zu ops %zu entries %zu\n", B.frames.size(), B.ops.size(), B.entries.size());
  for (auto& kv : ops_by_book)
    printf("  book %d (dim %d): %ld ops (%.2f per frame)\n", kv.first, S.books[kv.first].dimensions, kv.second,
           (double)kv.second / (double)B.frames.size());
  return 0;
}
