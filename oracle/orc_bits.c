/* oracle/orc_bits.c -- DataPacket bit reader + Utils (test infrastructure, see orc.h).
 *
 * DataPacket.cs:150-283 keeps a 64-bit bucket filled a byte at a time.  Its observable behaviour
 * is reproduced here with a plain bit cursor:
 *   - TryPeekBits(count) returns min(count, bitsRemaining) bits, zero-extended, and reports how
 *     many it got (DataPacket.cs:168-205: on ReadNextByte()==-1 it returns the bucket unmasked,
 *     which only ever holds the remaining bits).
 *   - SkipBits(count) advances; if fewer than `count` bits remain the cursor goes to the end and
 *     IsShort is set (DataPacket.cs:247-280); exactly `count` remaining does NOT set IsShort (:241-246).
 *   - ReadBits(count) = TryPeekBits + SkipBits (DataPacket.cs:150-160), so a read that runs off the
 *     end returns the zero-extended tail (quirk B-16).
 */
#include "orc_internal.h"

void orc_packet_init(orc_packet *p, const uint8_t *data, int len) {
  memset(p, 0, sizeof *p);
  p->data = data;
  p->len = len;
}

uint64_t orc_try_peek_bits(orc_packet *p, int count, int *bits_read) {
  int total = p->len * 8;
  int remaining = total - p->pos;
  int n, i;
  uint64_t v = 0;
  if (count <= 0) { /* DataPacket.cs:171-175 (count<0 / >64 throw; never requested by this path) */
    *bits_read = 0;
    return 0;
  }
  if (count > 64) count = 64;
  n = count < remaining ? count : remaining;
  if (n < 0) n = 0;
  for (i = 0; i < n;) {
    int bitpos = p->pos + i;
    int byte = p->data[bitpos >> 3];
    int sh = bitpos & 7;
    int take = 8 - sh;
    if (take > n - i) take = n - i;
    v |= (uint64_t)((byte >> sh) & ((1 << take) - 1)) << i;
    i += take;
  }
  *bits_read = n;
  return v;
}

void orc_skip_bits(orc_packet *p, int count) {
  int total = p->len * 8;
  if (count <= 0) return;
  if (total - p->pos >= count) {
    p->pos += count;
  } else {
    p->pos = total;
    p->is_short = 1;
  }
}

uint64_t orc_read_bits(orc_packet *p, int count) {
  int got;
  uint64_t v;
  if (count == 0) return 0; /* DataPacket.cs:152-153 */
  v = orc_try_peek_bits(p, count, &got);
  orc_skip_bits(p, count);
  return v;
}

int orc_read_bit(orc_packet *p) { /* Extensions.cs:59-62 */
  return orc_read_bits(p, 1) == 1;
}

/* ---- Utils.cs ---- */
int orc_ilog(int x) { /* Utils.cs:5-14 */
  int cnt = 0;
  while (x > 0) {
    ++cnt;
    x >>= 1;
  }
  return cnt;
}

uint32_t orc_bit_reverse(uint32_t n, int bits) { /* Utils.cs:21-28 */
  n = ((n & 0xAAAAAAAAu) >> 1) | ((n & 0x55555555u) << 1);
  n = ((n & 0xCCCCCCCCu) >> 2) | ((n & 0x33333333u) << 2);
  n = ((n & 0xF0F0F0F0u) >> 4) | ((n & 0x0F0F0F0Fu) << 4);
  n = ((n & 0xFF00FF00u) >> 8) | ((n & 0x00FF00FFu) << 8);
  n = (n >> 16) | (n << 16);
  /* C# masks shift counts of a 32-bit operand to 5 bits: x >> 32 == x >> 0 */
  return n >> ((32 - bits) & 31);
}

float orc_clip_value(float value, int *clipped) { /* Utils.cs:30-43 */
  if (value > .99999994f) {
    *clipped = 1;
    return 0.99999994f;
  }
  if (value < -.99999994f) {
    *clipped = 1;
    return -0.99999994f;
  }
  return value;
}

float orc_convert_from_vorbis_float32(uint32_t bits) { /* Utils.cs:45-59 */
  int32_t sign = ((int32_t)bits) >> 31; /* arithmetic: 0 or -1 */
  double exponent = (double)((int32_t)((bits & 0x7fe00000u) >> 21) - 788);
  /* uint ^ int promotes to long in C# */
  int64_t m = ((int64_t)(bits & 0x1fffffu) ^ (int64_t)sign) + (int64_t)(sign & 1);
  float mantissa = (float)m;
  return mantissa * (float)pow(2.0, exponent);
}
