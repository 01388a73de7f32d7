// host_bits.h -- LSB-first packet bit reader for the host parser.
//
// Mirrors the observable behaviour of NVorbis' DataPacket (DataPacket.cs:150-283): a peek past the
// end returns the zero-extended remaining bits and the count actually available (:168-205); a skip
// past the end parks the cursor at the end and raises IsShort (:247-280); ReadBits = peek + skip
// (:150-160).  Implemented as a bit cursor over the packet bytes (no bucket refill state).
#pragma once
#include <cstdint>
#include <cstring>

namespace nvh {

struct BitReader {
  const uint8_t* data = nullptr;
  int total_bits = 0;
  int pos = 0;
  bool is_short = false;

  BitReader() = default;
  BitReader(const uint8_t* d, int len_bytes) : data(d), total_bits(len_bytes * 8) {}

  inline uint64_t peek(int count, int* got) const {
    if (count <= 0) { *got = 0; return 0; }
    if (count > 64) count = 64;
    int remaining = total_bits - pos;
    int n = count < remaining ? count : remaining;
    int byte = pos >> 3, sh = pos & 7, filled = 0;
    if (n <= 57 && byte + 8 <= (total_bits >> 3)) {  // common case: one unaligned 64-bit load (little endian host)
      uint64_t w;
      std::memcpy(&w, data + byte, 8);
      *got = n;
      return n == 0 ? 0 : (w >> sh) & ((~0ull) >> (64 - n));
    }
    uint64_t v = 0;
    while (filled < n) {
      uint64_t b = (uint64_t)(data[byte++] >> sh);
      v |= b << filled;
      filled += 8 - sh;
      sh = 0;
    }
    if (n < 64) v &= (n == 0) ? 0 : ((~0ull) >> (64 - n));
    *got = n;
    return v;
  }
  inline void skip(int count) {
    if (count <= 0) return;
    if (total_bits - pos >= count) pos += count;
    else { pos = total_bits; is_short = true; }
  }
  inline uint64_t read(int count) {
    if (count == 0) return 0;
    int got;
    uint64_t v = peek(count, &got);
    skip(count);
    return v;
  }
  inline bool read_bit() { return read(1) == 1; }
};

}  // namespace nvh
