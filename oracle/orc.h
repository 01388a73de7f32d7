/*
 * oracle/orc.h -- CPU restatement of NVorbis' per-packet synthesis path (TEST INFRASTRUCTURE).
 *
 * This is the parity oracle: a plain-C, function-for-function restatement of the reference's
 * managed C# algorithm (reference paths cited as `File.cs:lines`, relative to
 * /root/reference/NVorbis/).  It is NOT the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (nvorbis_amd/) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" by any executable artefact of the reference -- the reference
 * is C# (no dotnet/mono in the build image) and ships no tests, golden PCM or KAT vectors.  The
 * oracle is pinned instead by (1) sample-count known answers derived from the shipped TestFiles
 * (final granule == emitted total), (2) the IMDCT closed-form identity, (3) window power
 * complementarity, (4) TDAC perfect reconstruction -- see tests/test_oracle_*.py.
 *
 * Float semantics follow .NET 6 RyuJIT x64: every float op rounds to binary32, no FMA
 * contraction, System.Math.* in double.  Build with -O2 -ffp-contract=off -fno-fast-math.
 */
#ifndef ORC_H
#define ORC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (the reference throws; the oracle records and returns) ---- */
enum {
  ORC_OK = 0,
  ORC_ERR_INVALID_DATA = -1, /* System.IO.InvalidDataException at header time        */
  ORC_ERR_ARGUMENT = -2,     /* ArgumentOutOfRangeException on Read()                */
  ORC_ERR_RUNTIME = -3,      /* IndexOutOfRange / NullReference / DivideByZero class */
  ORC_ERR_NOMEM = -4,
  ORC_ERR_NOT_VORBIS = -5,
  ORC_ERR_STATE = -6         /* InvalidOperationException (StreamDecoder.SeekTo)      */
};

/* ---- DataPacket (DataPacket.cs:150-283) ---- */
typedef struct orc_packet {
  const uint8_t *data;
  int len;      /* bytes */
  int pos;      /* bit cursor (== _readBits while not short) */
  int is_short; /* PacketFlags.IsShort */
  int is_resync;
  int is_eos;
  int has_granule;
  int64_t granule;
} orc_packet;

void orc_packet_init(orc_packet *p, const uint8_t *data, int len);
uint64_t orc_try_peek_bits(orc_packet *p, int count, int *bits_read);
void orc_skip_bits(orc_packet *p, int count);
uint64_t orc_read_bits(orc_packet *p, int count);
int orc_read_bit(orc_packet *p);

/* ---- stand-alone table builders / transforms (unit-testable) ---- */
/* Mdct.cs:13-21 -> MdctImpl.CalcReverse :65-313.  buf holds n floats; reads [0,n/2), writes [0,n). */
void orc_mdct_reverse(float *buf, int n);
/* Mdct.cs:30-63.  a[n/2], b[n/2], c[n/4], bitrev[n/8] */
void orc_mdct_tables(int n, float *a, float *b, float *c, uint16_t *bitrev);
/* Mode.cs:69-100 */
void orc_calc_window(int prev_block, int block, int next_block, float *out);
/* Mode.cs:102-117 */
void orc_calc_overlap(int prev_block, int block, int next_block, int *start, int *valid, int *total);
/* Utils.cs */
int orc_ilog(int x);
uint32_t orc_bit_reverse(uint32_t n, int bits);
float orc_convert_from_vorbis_float32(uint32_t bits);
/* Floor1.cs:345-410 table lookup (index must be 0..255) */
float orc_inverse_db(int i);
/* Floor1.cs:299-314 and :316-341 (render into v, multiplying) */
int orc_render_point(int x0, int y0, int x1, int y1, int X);
int orc_render_line_multi(int x0, int y0, int x1, int y1, float *v, int vlen);
/* Mapping.cs:137-182 inverse coupling of one (magnitude, angle) pair over cnt bins */
void orc_inverse_couple(float *magnitude, float *angle, int cnt);
/* Utils.cs:30-43 */
float orc_clip_value(float v, int *clipped);

/* ---- stream decoder (StreamDecoder.cs) over an in-memory packet list ---- */
typedef struct orc_decoder orc_decoder;

/* Ogg container -> decoder (minimal forward demux: Ogg/PageReaderBase.cs:33-70,227-292,
 * Ogg/PageReader.cs:27-93, Ogg/PacketProvider.cs:324-438, Ogg/Crc.cs).  First logical stream only. */
orc_decoder *orc_open_ogg(const uint8_t *bytes, size_t len, int *err);
/* The first logical stream as the reader for sources that cannot seek delivers it: ForwardOnlyPageReader.AddPage +
 * ForwardOnlyPacketProvider.GetPacket (Ogg/ForwardOnlyPageReader.cs:21-52, Ogg/ForwardOnlyPacketProvider.cs:36-67, 119-284).
 * Arrays are malloc'ed (caller frees); granule -1 = none (or a page value of -1); flags bit0 = EOS, bit1 = resync. */
int orc_ogg_demux_forward(const uint8_t *bytes, size_t len, uint8_t **out_bytes, int64_t **out_offs, int64_t **out_granule,
                          uint8_t **out_flags, int *out_n);
/* the seekable reader's list (Ogg/PageReader.cs, Ogg/StreamPageReader.cs:44-91, Ogg/PacketProvider.cs:324-438), same convention */
int orc_ogg_demux(const uint8_t *bytes, size_t len, uint8_t **out_bytes, int64_t **out_offs, int64_t **out_granule,
                  uint8_t **out_flags, int *out_n);
void orc_free(void *p);
/* IPacketProvider.SeekTo(granulePos, preRoll, GetPacketGranules) on the first logical stream of an Ogg file, for a reader
 * that has read every page (Ogg/PacketProvider.cs:56-295, Ogg/StreamPageReader.cs:122-264, StreamDecoder.cs:630-647; d supplies the
 * modes).  *packet_index = position in orc_ogg_demux's list of the packet GetNextPacket returns next, *granule_out = the method's
 * return value; ORC_ERR_ARGUMENT / ORC_ERR_INVALID_DATA / ORC_ERR_RUNTIME where the reference throws ArgumentOutOfRangeException /
 * InvalidDataException / faults on an index. */
int orc_ogg_seek(const uint8_t *bytes, size_t len, orc_decoder *d, int64_t granule_pos, int pre_roll, int64_t *packet_index,
                 int64_t *granule_out);
/* StreamDecoder.SeekTo(samplePosition, SeekOrigin.Begin) (StreamDecoder.cs:562-628) on a decoder opened with orc_open_ogg:
 * provider seek (orc_ogg_seek) one packet early, ResetDecoder, the pre-roll packet, the packet that holds the target, the
 * roll-forward.  ORC_ERR_STATE = InvalidOperationException; other codes as orc_ogg_seek.  (Where the roll-forward exceeds the
 * packet's output the managed Read spins; orc_read_samples then returns ORC_ERR_RUNTIME.) */
int orc_seek_to(orc_decoder *d, int64_t sample_position);
/* IPacketProvider.GetGranuleCount (Ogg/PacketProvider.cs:30-42) of the first logical stream: the largest page granule position */
int64_t orc_total_samples(const orc_decoder *d);
/* Raw packets: packet i = bytes[offs[i] .. offs[i+1]); granule[i] < 0 => none; flags bit0 = EOS, bit1 = resync.
 * The first three packets must be the Vorbis id / comment / setup headers. */
orc_decoder *orc_open_packets(const uint8_t *bytes, const int64_t *offs, const int64_t *granule,
                              const uint8_t *flags, int npackets, int *err);
void orc_close(orc_decoder *d);

int orc_channels(const orc_decoder *d);
int orc_sample_rate(const orc_decoder *d);
int orc_block0(const orc_decoder *d);
int orc_block1(const orc_decoder *d);
int orc_packet_count(const orc_decoder *d); /* audio + header packets in the list */
void orc_set_clip_samples(orc_decoder *d, int on); /* StreamDecoder.ClipSamples, default on */
int orc_has_clipped(const orc_decoder *d);
int orc_is_end_of_stream(const orc_decoder *d);
int64_t orc_sample_position(const orc_decoder *d);
int orc_last_error(const orc_decoder *d);

/* VorbisReader.ReadSamples(float[],int,int) (VorbisReader.cs:336-345) -> StreamDecoder.Read (:320-389).
 * Returns floats written (>= 0) or a negative ORC_ERR_*. */
int orc_read_samples(orc_decoder *d, float *buffer, int buffer_len, int offset, int count);

/* Per-packet trace hook for tests: after each ReadNextPacket the oracle appends
 * (start, valid, total, decoded_ok) here if tracing is on (StreamDecoder.cs:417-463). */
typedef struct orc_frame_trace {
  int32_t start, valid, total, ok, block_size, window_index;
} orc_frame_trace;
void orc_enable_trace(orc_decoder *d, int on);
int orc_trace_count(const orc_decoder *d);
const orc_frame_trace *orc_trace_data(const orc_decoder *d);

/* Decode ONE audio packet's Mode.Decode (Mode.cs:153-170) into caller planes [ch][block1], no overlap.
 * Used to compare the product's per-frame synthesis (windowed blocks) with the oracle.
 * Returns 1 if decoded, 0 if the packet was rejected, <0 on error. */
int orc_decode_packet_block(orc_decoder *d, const uint8_t *pkt, int len, float *planes /* ch*block1 */,
                            int *start, int *valid, int *total, int *block_size);

/* IFloor.Apply (Floor1.cs:186-222) of the stream's floor `floor_index` on raw posts as Floor1.Unpack leaves them:
 * residue holds block_size/2 values (of a block1-long buffer in the reference; reslen says how long).
 * Returns 0, or ORC_ERR_RUNTIME where the reference would throw (residue is then partly modified). */
int orc_floor1_apply_posts(orc_decoder *d, int floor_index, int block_size, const int *posts, int post_count,
                           float *residue, int reslen);
/* IResidue.Decode (Residue0.cs:119-178 and the WriteVectors of types 0/1/2) of residue `residue_index` reading `pkt`
 * from bit `bit_offset`, adding into planes [channels][block1]; *bits_consumed = cursor movement. */
int orc_residue_decode_at(orc_decoder *d, int residue_index, const uint8_t *pkt, int len, int bit_offset,
                          int any_channel_decodes, int block_size, float *planes, int *bits_consumed);
/* The IResidue.Decode calls Mapping.DecodePacket made for the last packet given to orc_decode_packet_block:
 * returns their number; pos[i] = packet cursor before call i, idx[i] = residue index, *any = some channel executes. */
int orc_last_residue_calls(const orc_decoder *d, int *pos, int *idx, int cap, int *any);
/* block flag, block size and mapping of mode `mode_index`; returns the number of modes. */
int orc_mode_info(const orc_decoder *d, int mode_index, int *block_flag, int *block_size, int *mapping);
/* IFloor.Apply (Floor0.cs:152-212) of floor `floor_index` for Data.Amp = amp and Data.Coeff = coeff[0..order). */
int orc_floor0_apply_coeffs(orc_decoder *d, int floor_index, int block_size, float amp, const float *coeff,
                            float *residue, int reslen);
/* type, post count (Floor1 _xList.Length; Floor0 _order) and _range of floor `floor_index`; returns the number of floors. */
/* test hook: the tables Codebook.Init / Huffman.GenerateTable built (Codebook.cs:59-283, Huffman.cs:15-76), same calling
 * convention as the product's nvh_stream_codebook_info / nvh_stream_codebook_tables */
int orc_codebook_info(const orc_decoder *d, int book_index, int *dimensions, int *entries, int *map_type, int *prefix_bits,
                      int *max_bits, int *n_prefix, int *n_overflow);
int orc_codebook_tables(const orc_decoder *d, int book_index, int32_t *lengths, float *lookup, int32_t *prefix, int32_t *overflow);
int orc_book_count(const orc_decoder *d);
int orc_floor_info(const orc_decoder *d, int floor_index, int *type, int *post_count, int *range);

/* IFloorData of one channel as Mapping.DecodePacket left it for the last packet given to orc_decode_packet_block
 * (returns the floor type, or a negative ORC_ERR_*), and the static structure of a mapping. */
int orc_last_floor_data(const orc_decoder *d, int channel, int *execute, int *posts, int *post_count, float *amp, float *coeff,
                        int coeff_cap);
int orc_mapping_info(const orc_decoder *d, int mapping_index, int *coupling_steps, int *magnitude, int *angle, int cap,
                     int *channel_floor, int ch_cap);

/* Test instrumentation: residue coverage of the calling thread's decodes between begin and end.
 * mask[channel * plane_len + bin] gets bit s set when cascade stage s added a value to that bin. */
int orc_coverage_begin(int channels, int plane_len);
int orc_coverage_end(unsigned char *mask_out);

#ifdef __cplusplus
}
#endif
#endif /* ORC_H */
