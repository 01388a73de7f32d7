/*
 * nvorbis_hip.h -- C ABI of libnvorbis_hip.so, the MI355X (gfx950) back end for NVorbis'
 * per-packet synthesis path.
 *
 * The reference (NVorbis 0.10.5, managed C#) has no native boundary; its synthesis plug-ins sit
 * behind the internal interfaces IMdct / IFloor / IResidue / IMapping / IMode created by IFactory
 * (NVorbis/Contracts/I*.cs, NVorbis/Factory.cs:5-59) and are driven once per packet by
 * StreamDecoder.DecodeNextPacket (NVorbis/StreamDecoder.cs:497-506).  This header declares what a
 * P/Invoke binding for that path binds to (see INTEGRATION.md for the C# side):
 *
 *   level 1  batched, device-pointer entry points that mirror the interface methods one to one
 *            (unit parity tests, GpuMdct / GpuFloor / GpuResidue / GpuMapping shims);
 *   level 2  a stream object that takes raw Vorbis packets exactly as IPacketProvider hands them
 *            to StreamDecoder (bytes + granule position + end-of-stream / resync flags), parses
 *            them on the host, synthesises whole batches of frames on the GPU and returns
 *            interleaved float PCM with VorbisReader.ReadSamples semantics.
 *
 * Conventions: every function returns an int status (NVH_OK == 0, negative = error, never throws or
 * unwinds across the ABI); handles are opaque; `d_` pointers are device (HBM) addresses, all other
 * pointers are host addresses owned by the caller for the duration of the call.  A handle must not
 * be used from two threads at once; distinct handles are independent.
 */
#ifndef NVORBIS_HIP_H
#define NVORBIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (C# shim maps them back to the reference's exception types) ---- */
#define NVH_OK 0
#define NVH_ERR_INVALID_DATA (-1) /* System.IO.InvalidDataException (Mapping.cs:40-77, Floor1.cs:122, Residue0.cs:64,73, Codebook.cs:63,161, Factory.cs:29,38,56) */
#define NVH_ERR_ARGUMENT (-2)     /* ArgumentOutOfRangeException / ArgumentException (StreamDecoder.cs:322-325, Floor1.cs:188) */
#define NVH_ERR_RUNTIME (-3)      /* IndexOutOfRange / NullReference / DivideByZero class faults of the managed code */
#define NVH_ERR_NOMEM (-4)
#define NVH_ERR_NOT_VORBIS (-5)   /* header signature mismatch (StreamDecoder.cs:145-155) */
#define NVH_ERR_DEVICE (-6)       /* HIP runtime error; nvh_last_hip_error() has the code */
#define NVH_ERR_UNSUPPORTED (-7)  /* legal stream outside the documented limits of this build */
#define NVH_ERR_NO_GPU (-8)       /* no gfx950 device visible: there is NO CPU fallback */

typedef struct nvh_ctx nvh_ctx;       /* device + HIP stream + per-n IMDCT table cache */
typedef struct nvh_stream nvh_stream; /* one logical Vorbis stream: setup tables in HBM + overlap state */
typedef struct nvh_batch nvh_batch;   /* one parsed batch of frames resident in HBM */

/* packet flags, as IPacket exposes them (Contracts/IPacket.cs: IsEndOfStream, IsResync) */
#define NVH_PKT_EOS 1
#define NVH_PKT_RESYNC 2

const char *nvh_version(void);
int nvh_last_hip_error(void);
int nvh_device_count(void);

/* ---- context ---- */
int nvh_ctx_create(int device, nvh_ctx **out);
void nvh_ctx_destroy(nvh_ctx *ctx);
/* Launch on an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream. */
int nvh_ctx_set_hip_stream(nvh_ctx *ctx, void *hip_stream);
int nvh_ctx_synchronize(nvh_ctx *ctx);
/* Launch shape of the GPU packet parser (nvh_stream_set_gpu_parse) for the streams of this context: packets per wavefront,
 * a power of two 1..64, 0 = automatic (one packet per wavefront up to 4096 packets per batch: the lowest latency for a lone
 * stream).  A host that keeps many contexts busy at once -- a corpus worker pool, one context per thread -- gives each
 * of them 32 (an upper limit for batches below a thousand packets, which keep at least 256 wavefronts: short files parse
 * one packet per wavefront all the same): where packets share wavefronts every lane walks its own packet (k_parse_slab_f;
 * a wavefront costs the same at 8 or 64 packets), a parse is a few dozen wavefronts, and the parses of all workers fit the
 * chip side by side (the corpus of BASELINE configs[4]: decode pass 0.82 -> 0.51 s with 8 packets per wavefront and 16
 * workers in round 5, 0.35 s with 32 and 32; with the process started under GPU_MAX_HW_QUEUES=16, the HIP runtime's
 * default of 4 hardware queues lets only four kernels run at once).
 * No counterpart in the reference (its decoder is one thread per stream); results do not depend on it. */
int nvh_ctx_set_parse_lanes(nvh_ctx *ctx, int lanes);

/* ---- level 1: batched mirrors of the plug-in interface methods (device pointers) ----
 * nvh_inverse_couple, nvh_mdct_reverse, nvh_window_apply and nvh_overlap_buffers only enqueue their kernel on the
 * context's stream (nvh_ctx_synchronize, or work queued on the same HIP stream, orders against them); the entry points
 * that return per-item status or read packets (nvh_floor*_apply, nvh_residue_decode, nvh_copy_buffer, nvh_mode_decode)
 * are synchronous. */

/* One inverse square-polar coupling step over two device vectors of `count` floats, in place (Mapping.cs:150-178). */
int nvh_inverse_couple(nvh_ctx *c, float *d_magnitude, float *d_angle, int count);

/* IFloor.Apply(IFloorData, int blockSize, float[] residue) for a Floor1 (Contracts/IFloor.cs, Floor1.cs:186-341:
 * UnwrapPosts, the walk over the sorted posts, RenderLineMulti) of stream `s`, on `batch` device vectors:
 * item b multiplies d_residue[b*stride .. +block_size/2) by the curve its posts describe, or clears it when
 * post_counts[b] == 0 (:218-221).  posts: host, [batch][64] raw values as Floor1.Unpack leaves them in
 * Data.Posts (:135-184), each 0..65535; post_counts[b] is 0 or the floor's post count (nvh_stream_floor_info).
 * status (host, [batch], may be NULL): NVH_OK, or NVH_ERR_RUNTIME where the reference would index
 * inverse_dB_table out of range (the item's vector is then unspecified, as after the managed exception);
 * with status == NULL the first such code is the return value.  Synchronous. */
int nvh_floor1_apply(nvh_stream *s, int floor_index, int block_size, int batch, const int32_t *posts,
                     const int32_t *post_counts, float *d_residue, int64_t stride, int32_t *status);
/* IFloor.Apply for a Floor0 (Floor0.cs:152-212): item b scales d_residue[b*stride .. +block_size/2) by the curve of
 * its LSP coefficients (coeffs: host, [batch][coeff_stride], the first `order` of each row as Floor0.Unpack leaves
 * them in Data.Coeff, :98-150) and amplitude amps[b] (Data.Amp), or clears it when amps[b] <= 0 (:208-211).
 * Floating point: the curve's value per Bark section (cos / sqrt / exp in double, rounded to float, Floor0.cs:167,198,201) is
 * evaluated on the calling thread with the host's math library and only gathered and multiplied on the device; values
 * differ from the managed reference only where the two platforms' libm differ in the last ulp of a double.  status as for
 * nvh_floor1_apply. */
int nvh_floor0_apply(nvh_stream *s, int floor_index, int block_size, int batch, const float *amps, const float *coeffs,
                     int coeff_stride, float *d_residue, int64_t stride, int32_t *status);
/* type (0/1), number of posts (Floor1: _xList.Length, Floor1.cs:93-107; Floor0: _order) and _range (Floor1.cs:76)
 * of floor `floor_index`. */
int nvh_stream_floor_info(const nvh_stream *s, int floor_index, int *type, int *post_count, int *range);

/* Mode.cs:24-50 of mode `mode_index`: its block flag, block size and mapping index (NVH_ERR_ARGUMENT past the last mode). */
int nvh_stream_mode_info(const nvh_stream *s, int mode_index, int *block_flag, int *block_size, int *mapping);
/* ICodebook (Contracts/ICodebook.cs:5-13) of the stream's codebook `book_index`, as Codebook.Init built it
 * (Codebook.cs:59-283) with Huffman.GenerateTable's decode tables (Huffman.cs:15-76): Dimensions, Entries, MapType, and the
 * sizes of the tables nvh_stream_codebook_tables copies out.  n_prefix = 1 << prefix_bits slots (0: the tree was never
 * built), n_overflow = -1 for the reference's null overflow list. */
int nvh_stream_codebook_info(const nvh_stream *s, int book_index, int *dimensions, int *entries, int *map_type, int *prefix_bits,
                             int *max_bits, int *n_prefix, int *n_overflow);
/* lengths[entries] (Codebook.cs:76-160; -1 = unused entry), lookup[entries * dimensions] = the indexer's table
 * (Codebook.cs:222-283, :322; untouched for map type 0), prefix[n_prefix * 5] and overflow[n_overflow * 5] = the Huffman nodes
 * as (present, value, length, bits, mask) (Huffman.cs:15-76).  Any pointer may be NULL. */
int nvh_stream_codebook_tables(const nvh_stream *s, int book_index, int32_t *lengths, float *lookup, int32_t *prefix,
                               int32_t *overflow);

/* IResidue.Decode(IPacket, bool[] doNotDecodeChannel, int blockSize, float[][] buffer) (Contracts/IResidue.cs:6;
 * Residue0.cs:119-201, Residue1.cs:8-26, Residue2.cs:10-47) for residue `residue_index` of stream `s`: reads the
 * classifications and codebook entries from `pkt` starting at bit `bit_offset` and adds the decoded vectors into
 * d_buffer, device planes [channels][block1] (the reference's float[channels][block1Size] working buffer).
 * any_channel_decodes = doNotDecodeChannel contains a false (:125; otherwise the call reads nothing).
 * *bits_consumed: how far the packet cursor moved.  The stream must have nothing pending.  Synchronous. */
int nvh_residue_decode(nvh_stream *s, int residue_index, const uint8_t *pkt, int len, int bit_offset,
                       int any_channel_decodes, int block_size, float *d_buffer, int *bits_consumed);

/* Mode.Decode's window loop (Mode.cs:160-166) for mode `mode_index`: d_buf[b*stride + i] *= window[i], i < the
 * mode's block size, b < batch; the window is the one the packet's previous/next flag bits select (Mode.cs:135). */
int nvh_window_apply(nvh_stream *s, int mode_index, int prev_flag, int next_flag, int batch, float *d_buf,
                     int64_t stride);
/* StreamDecoder.OverlapBuffers (StreamDecoder.cs:532-541) on device planes [channels][plane_stride]:
 * next[c][next_start + j] += previous[c][prev_start + j] for j < prev_stop - prev_start. */
int nvh_overlap_buffers(nvh_ctx *c, const float *d_previous, float *d_next, int prev_start, int prev_stop,
                        int next_start, int channels, int64_t plane_stride);
/* ClippingCopyBuffer / CopyBuffer (StreamDecoder.cs:391-415, Utils.ClipValue Utils.cs:30-43): `count` samples per
 * channel from index `start` of device planes [channels][plane_stride], interleaved into d_target[count*channels];
 * clip != 0 clamps to +-0.99999994f and reports in *clipped whether any value was (HasClipped).  Synchronous. */
int nvh_copy_buffer(nvh_ctx *c, const float *d_planes, int start, int count, int channels, int64_t plane_stride,
                    float *d_target, int clip, int *clipped);

/* Device memory for callers without a HIP binding of their own (csharp/GpuFactory.cs stages the managed float[] arrays of
 * the plug-in interfaces through these).  Copies complete before the call returns. */
int nvh_dev_alloc(nvh_ctx *c, size_t bytes, void **out);
void nvh_dev_free(nvh_ctx *c, void *d_ptr);
int nvh_dev_upload(nvh_ctx *c, void *d_dst, const void *h_src, size_t bytes);
int nvh_dev_download(nvh_ctx *c, void *h_dst, const void *d_src, size_t bytes);

/* Measurement helper (no reference counterpart): `iters` passes of a float4 copy kernel over `bytes` bytes (a multiple
 * of 16) from d_src to d_dst, timed with HIP events on the context's stream; *ms = total.  bench.py reports
 * 2 * bytes * iters / ms as the measured HBM ceiling next to the roofline figure. */
int nvh_measure_copy(nvh_ctx *c, const void *d_src, void *d_dst, size_t bytes, int iters, float *ms);

/* IMdct.Reverse(float[] samples, int sampleCount) (Contracts/IMdct.cs:5, Mdct.cs:13-21) on `batch`
 * buffers: buffer b = d_buf + b*stride holds n floats, reads [0,n/2), writes [0,n).  n = 64..8192. */
int nvh_mdct_reverse(nvh_ctx *ctx, int n, int batch, float *d_buf, int64_t stride);

/* Table builders (host, exact reference typing): Mdct.cs:30-63, Mode.cs:69-117. */
int nvh_mdct_tables(int n, float *a /*n/2*/, float *b /*n/2*/, float *c /*n/4*/, uint16_t *bitrev /*n/8*/);
int nvh_calc_window(int prev_block, int block, int next_block, float *out /*block*/);
int nvh_calc_overlap(int prev_block, int block, int next_block, int *start, int *valid, int *total);

/* ---- level 2: stream ---- */

/* StreamDecoder..ctor -> ProcessHeaderPackets (StreamDecoder.cs:50-127): the three Vorbis header
 * packets.  Builds every table of LoadBooks (:226-289) and uploads it once.  ctx == NULL creates a
 * host-only stream that can parse packets but returns NVH_ERR_NO_GPU from every synthesis call. */
int nvh_stream_open(nvh_ctx *ctx, const uint8_t *id_pkt, int id_len, const uint8_t *comment_pkt, int comment_len,
                    const uint8_t *setup_pkt, int setup_len, nvh_stream **out);
void nvh_stream_close(nvh_stream *s);
/* IStreamDecoder.UpperBitrate / NominalBitrate / LowerBitrate (Contracts/IStreamDecoder.cs; the three 32-bit fields
 * of the identification header, StreamDecoder.cs:191-193). */
int nvh_stream_bitrates(const nvh_stream *s, int *upper, int *nominal, int *lower);
int nvh_stream_info(const nvh_stream *s, int *channels, int *sample_rate, int *block0, int *block1);
/* IStreamDecoder.ClipSamples (StreamDecoder.cs:723, default on) / HasClipped (:728) */
/* Page-locked host memory for PCM destinations: nvh_stream_synth writes a pinned pcm_host directly with the copy
 * engine (no bounce buffer, no memcpy on the calling thread).  Any other pinned allocation works as well. */
int nvh_pinned_alloc(size_t bytes, void **out);
void nvh_pinned_free(void *p);
int nvh_stream_set_clip(nvh_stream *s, int on);
/* Parse audio packets on the GPU (kernels_parse.hip: floors, residue classification and VQ entry decode, one lane per
 * packet) instead of on the calling thread; the host then only reads each packet's mode number and window flags.
 * Same PCM.  Difference in error behaviour: a packet the managed decoder would have thrown on (NVH_ERR_RUNTIME) is
 * reported by nvh_stream_synth for the whole look-ahead batch instead of by nvh_stream_push_packet for that packet.
 * NVH_ERR_UNSUPPORTED for stream shapes outside the GPU parser's limits (Floor0, > 8 channels); call between batches.
 * The environment variable NVH_GPU_PARSE=1 turns it on for every eligible stream. */
int nvh_stream_set_gpu_parse(nvh_stream *s, int on);
int nvh_stream_has_clipped(nvh_stream *s, int *clipped);
/* IStreamDecoder.SamplePosition after everything parsed so far has been read (StreamDecoder.cs:718) */
int nvh_stream_position(const nvh_stream *s, int64_t *position, int64_t *emitted, int *eos);

/* Host parse of one audio packet into the pending batch (DecodeNextPacket, StreamDecoder.cs:465-530,
 * bit-consuming half).  granule < 0 = packet carries no granule position. */
int nvh_stream_push_packet(nvh_stream *s, const uint8_t *data, int len, int64_t granule, int flags);
/* nvh_stream_push_packet over a packet array in one call (packet i = bytes[offsets[i], offsets[i+1]); granules / flags
 * may be NULL): the look-ahead loop of a batched caller without one FFI transition per packet.  Stops after
 * max_packets, at the first error, or once the stream has seen its end-of-stream packet; *consumed = packets taken. */
int nvh_stream_push_packets(nvh_stream *s, const uint8_t *bytes, const int64_t *offsets, const int64_t *granules,
                            const uint8_t *flags, int n, int max_packets, int *consumed);
/* _hasPosition / _currentPosition (StreamDecoder.cs:35-39) after everything pushed so far was read.  Setting them is
 * what a caller does that starts a decoder in the middle of a stream (the state SeekTo leaves behind, :562-628):
 * nvorbis_amd/corpus.py's chunked decode pushes one lead-in packet, then sets the state the serial decoder has there. */
int nvh_stream_position_state(const nvh_stream *s, int *has_position, int64_t *position);
int nvh_stream_set_position_state(nvh_stream *s, int has_position, int64_t position);
/* Index of a run of `n` audio packets (packet i = bytes[offsets[i] .. offsets[i+1]), granule -1 = none, flags NVH_PKT_*;
 * granules / flags may be NULL) as a serial decoder that starts with packet 0 of the run sees them -- the integer half of
 * ReadNextPacket (StreamDecoder.cs:417-463) and Mode.GetPacketInfo (Mode.cs:119-151) only, no floor / residue bits:
 *   position_after[i]  _currentPosition once everything packet i lets the decoder emit was read
 *   emitted_after[i]   samples per channel emitted so far
 *   state_after[i]     bit 0: the packet decodes; bit 1: and its overlap stays out of its own tail (a decoder may start
 *                      right after it with this packet as its only lead-in); bit 2: _hasPosition; bit 3: _eosFound
 * *total_emitted: samples per channel the whole run yields, including the drain of the last block's tail when the run
 * ends without an end-of-stream packet (StreamDecoder.cs:352-356).  Any output pointer may be NULL.
 * Host only; does not touch the stream's own state.  Callers: seeking and the chunked decode of nvorbis_amd. */
int nvh_stream_index_packets(const nvh_stream *s, const uint8_t *bytes, const int64_t *offsets, const int64_t *granules,
                             const uint8_t *flags, int n, int64_t *position_after, int64_t *emitted_after,
                             uint8_t *state_after, int64_t *total_emitted);
/* StreamDecoder.GetPacketGranules (StreamDecoder.cs:630-647): the sample count a packet stands for in the reference's
 * page granule arithmetic (Mode.GetPacketSampleCount, Mode.cs:172-176); 0 for resync, non-audio and short packets and
 * for an invalid mode number.  What a binding passes to IPacketProvider.SeekTo as its GetPacketGranuleCount delegate. */
int nvh_stream_packet_sample_count(const nvh_stream *s, const uint8_t *pkt, int len, int is_resync, int *count);
/* ResetDecoder (StreamDecoder.cs:295-305): previous block, position, end-of-stream and clipped flag forgotten, pending
 * frames dropped; the next packet pushed is a first packet again (emits nothing, :446-450).  SeekTo = reset, push the
 * pre-roll packet, nvh_stream_set_position_state(1, position of the packet that follows), carry on, discard the
 * roll-forward samples (:596-627). */
int nvh_stream_reset(nvh_stream *s);
/* Forget the pending (parsed, not yet synthesised) frames without synthesising them: for callers that only want the
 * integer geometry (positions, sample counts) of a stretch of packets. */
int nvh_stream_drop_pending(nvh_stream *s);
/* The packet provider returned null (StreamDecoder.cs:472-475). */
int nvh_stream_push_end(nvh_stream *s);
/* Pending (parsed, not yet synthesised) work. */
/* Frame geometry of the pending batch, 8 int32 per frame: block size, start, valid, total
 * (Mode.GetPacketInfo, Mode.cs:119-151, valid after the EOS trim of StreamDecoder.cs:429-437),
 * emit_start, emit_count, overlap source frame (-1 none, -2 carried tail), overlap length. */
int nvh_stream_pending_geometry(const nvh_stream *s, int32_t *out, int cap_frames);
int nvh_stream_pending(const nvh_stream *s, int *frames, int64_t *pcm_samples_per_channel);
/* The pending frames in the form the synthesis kernels fetch (per-frame slabs: the integer half of Floor1.Apply --
 * UnwrapPosts and the walk over the sorted posts, Floor1.cs:196-297 -- as line segments, the vector writes of
 * Residue0.cs:132-175 / Residue2.cs:23-47 as chain-major records; the entry section in DIGIT form for setups whose residue books
 * have at most 63 lattice values (nvh_format.h: NVH_SLAB_RGEOM_DIGITS -- one byte per vector component, the byte offset of its
 * float from the book's first word in the value pool; a record's x then counts 2-byte units of that section, its y holds the
 * book's value-pool offset), else as uint16 entry numbers), written by the host parser's thread.  Host only, for
 * tests and tools: buf receives the slabs back to back, first_unit[f] the first 16-byte unit of frame f's slab
 * (first_unit[frames] = total units).  NVH_ERR_UNSUPPORTED: the stream shape is outside the slab kernels' contract.
 * *bytes is set even when cap is too small (NVH_ERR_ARGUMENT then). */
int nvh_stream_pending_slabs(const nvh_stream *s, uint8_t *buf, int64_t cap, int64_t *bytes, uint32_t *first_unit,
                             int cap_frames);
/* The setup's lattice pool as the synthesis kernels hold it: per lattice codebook (Codebook.cs:222-283, lookup type 1 without
 * sequence_p) its distinct component values as float bits, then the reciprocals ceil(2^32 / lat_values^i) for i < dimensions;
 * a slab record of the ENTRY form points at a book's first value there.  Behind it -- only for setups that take the digit form --
 * the VALUE pool: per book its distinct values again, then +0.0f (where the bytes of a vector that was never added point); a slab
 * record of the DIGIT form holds the offset of the book's first value-pool word, counted from the lattice pool's start.  Host only,
 * for tests and tools.  *words is set even when cap_words is too small (NVH_ERR_ARGUMENT then). */
int nvh_stream_lattice_pool(const nvh_stream *s, uint32_t *out, int64_t cap_words, int64_t *words);
/* The setup's VQ table pool (Codebook.cs:222-283: every book's lookup table, entries x dimensions floats, book after book).  A
 * slab record that names a book with an EXPLICIT table (lookup type 2, or type 1 with sequence_p: lat_values = 0 in the record)
 * points at one word of the lattice pool, that book's offset in this pool; component d of entry e is float offset + e * dim + d.
 * Host only, for tests and tools.  *floats is set even when cap_floats is too small (NVH_ERR_ARGUMENT then). */
int nvh_stream_vq_pool(const nvh_stream *s, float *out, int64_t cap_floats, int64_t *floats);

/* Synthesise the pending batch: H2D descriptors -> kernels -> interleaved PCM.  Exactly one of
 * pcm_host / d_pcm is non-NULL; capacity is in floats and must hold pending samples * channels.
 * Advances the overlap state (the last block's tail is carried to the next batch). */
int nvh_stream_synth(nvh_stream *s, float *pcm_host, float *d_pcm, int64_t capacity, int64_t *written);
/* Pipelined form for a destination in page-locked host memory (nvh_pinned_alloc): nvh_stream_synth_begin queues the upload, the
 * synthesis and -- on a copy stream of its own -- the transfer of the PCM and returns (*expected = floats the batch will
 * deliver); nvh_stream_synth_end waits for the OLDEST outstanding batch and reports what nvh_stream_synth would have (error
 * codes, *written, nvh_stream_parse_errors).  Up to two batches may be outstanding: the PCIe transfer of batch i overlaps the
 * upload, the parse and the kernels of batch i+1 (begin itself first waits for the kernels of the batch before it -- both flights
 * share one scratch batch -- so what overlaps batch i's kernels is the pushing of batch i+1, which happens before its begin);
 * each needs its own destination buffer until its end call returns.  No counterpart in the
 * reference (its Read is synchronous); nvh_stream_synth must not be mixed in while batches are outstanding (NVH_ERR_ARGUMENT). */
int nvh_stream_synth_begin(nvh_stream *s, float *pcm_host, int64_t capacity, int64_t *expected);
int nvh_stream_synth_end(nvh_stream *s, int64_t *written);
/* After nvh_stream_synth returned an error code together with *written > 0 (GPU-parse mode: packets of the batch made
 * the parser fail -- with the codes nvh_stream_push_packet returns for them in host-parse mode -- and the batch was
 * parsed again on the host without them): every such packet in stream order, codes[i] and samples_before[i] = the
 * samples per channel of that batch's PCM that precede it, i.e. where the reference's exception would have surfaced.
 * *count = how many there are (0 when the last synthesis reported none); at most `cap` are written. */
int nvh_stream_parse_errors(const nvh_stream *s, int32_t *codes, int64_t *samples_before, int cap, int *count);

/* ---- device-resident batches (benchmarks, pipelined callers) ---- */
/* Move the pending batch into HBM as an object of its own; the stream's pending batch becomes empty
 * and its overlap state advances as if the batch had been synthesised. */
/* IMode.Decode (Mode.cs:153-170) on ONE audio packet: the bit-consuming half on the host, then floors, residue adds, inverse
 * coupling, floor apply, IMDCT and window on the GPU -- the windowed block before any overlap -- into d_block
 * [channels][block1] (device).  Independent of the stream's decode state (needs an empty pending batch); *decoded = 0
 * when the reference returns without decoding the packet.  Geometry as Mode.GetPacketInfo reports it. */
int nvh_mode_decode(nvh_stream *s, const uint8_t *pkt, int len, float *d_block, int *decoded, int *block_size, int *start,
                    int *valid, int *total);

int nvh_batch_upload(nvh_stream *s, nvh_batch **out);
int nvh_batch_info(const nvh_batch *b, int *frames, int *chan_frames, int64_t *pcm_samples_per_channel,
                   int64_t *descriptor_bytes);
/* Descriptor element counts: frames, channel-frames, residue passes, residue ops, VQ entries, floor1 posts,
 * floor0 coefficients, nanoseconds the descriptor -> slab conversion of this upload took on the GPU (0: none). */
int nvh_batch_stats(const nvh_batch *b, int64_t *out8);
/* Names of the kernels behind the four timing slots of nvh_batch_time, as launched last (comma separated,
 * "-" = empty slot): which of the kernel variants ran depends on the stream shape. */
int nvh_batch_kernels(const nvh_batch *b, char *buf, int cap);
/* The same for the batch nvh_stream_synth / nvh_stream_synth_begin launched last (the stream's own look-ahead batch). */
int nvh_stream_kernels(const nvh_stream *s, char *buf, int cap);
/* Launch the synthesis kernels for a resident batch (asynchronous on the context's stream);
 * may be repeated, results are identical each time.  d_pcm holds samples*channels floats. */
int nvh_batch_synth(nvh_batch *b, float *d_pcm, int64_t capacity);
/* Time `iters` repetitions with hipEvents on the launch stream: total milliseconds for the whole
 * pipeline, and per timing slot (spectrum: residue | couple+floor, or fused in slot 1; imdct+window; overlap+emit;
 * see nvh_batch_kernels).  A slot brackets its launches with event records, which costs ~2 us per slot. */
int nvh_batch_time(nvh_batch *b, float *d_pcm, int64_t capacity, int iters, float *total_ms, float *kernel_ms /*[4]*/);
void nvh_batch_free(nvh_batch *b);

/* ---- container helper (minimal forward-only demux so .ogg files can feed the path) ----
 * Splits the first logical stream of an Ogg file into packets the way NVorbis' seekable reader
 * delivers them (Ogg/PageReader.cs:27-93, Ogg/PacketProvider.cs:324-438).  Call with NULL outputs to
 * size, then again with buffers: the sizing call keeps its result for the fill call that follows on the same
 * thread with the same bytes (the file is demultiplexed once; any other call sequence simply demultiplexes again). */
int nvh_ogg_demux(const uint8_t *bytes, size_t len, uint8_t *pkt_bytes, int64_t pkt_bytes_cap, int64_t *offsets,
                  int64_t *granules, uint8_t *flags, int pkt_cap, int *npackets, int64_t *total_bytes);
/* The same for logical stream `stream_index` of a multiplexed or chained file; *nstreams (may be NULL) = number of
 * logical streams in the file.  A stream begins with the first page of a serial number that has no open stream, ends with
 * its end-of-stream page (a later page with the same serial number starts another one); a page without packets is refused
 * and its serial number ignored from then on (Ogg/PageReader.cs:126-158, Ogg/PageReaderBase.cs:72-85).  Streams that are
 * not Vorbis are listed too: nvh_stream_open refuses them (NVH_ERR_NOT_VORBIS), as VorbisReader's new-stream callback
 * does (VorbisReader.cs:74-87).  NVH_ERR_ARGUMENT for a stream_index past the last stream. */
int nvh_ogg_demux_stream(const uint8_t *bytes, size_t len, int stream_index, uint8_t *pkt_bytes, int64_t pkt_bytes_cap,
                         int64_t *offsets, int64_t *granules, uint8_t *flags, int pkt_cap, int *npackets,
                         int64_t *total_bytes, int *nstreams);

/* The index form of nvh_ogg_demux_stream, for a pass that only asks what the stream decodes to (a corpus transcoder sizing its
 * output before the first kernel runs): the same page walk and packet rules (Ogg/PageReaderBase.cs:227-292 header + lacing
 * values, Ogg/PacketProvider.cs:324-438) WITHOUT the page checksums and without the packets' bodies.  The first three packets
 * (the Vorbis headers) are delivered whole, every other packet as its first (up to) 8 bytes -- packet type, mode number and window
 * flags, all nvh_stream_index_packets reads, are in the first two; *payload_bytes = the bytes the packets really have.  One call,
 * caller's buffers (pkt_bytes_cap >= len and pkt_cap >= len / 27 + 8 always suffice); NVH_ERR_ARGUMENT when they are too small.
 * A damaged page passes unnoticed here: the pass that decodes the file demultiplexes it with the checksums and compares the
 * packet count and the payload size with this call's (a refused page changes both). */
int nvh_ogg_index_packets(const uint8_t *bytes, size_t len, int stream_index, uint8_t *pkt_bytes, int64_t pkt_bytes_cap,
                          int64_t *offsets, int64_t *granules, uint8_t *flags, int pkt_cap, int *npackets,
                          int64_t *total_bytes, int64_t *payload_bytes, int *nstreams);

/* The packet list of a source that cannot seek: ForwardOnlyPageReader + ForwardOnlyPacketProvider
 * (Ogg/ForwardOnlyPageReader.cs:21-52, Ogg/ForwardOnlyPacketProvider.cs:36-67, 119-290), same calling convention.  It differs from
 * the seekable reader's list in the resync marks (the first page always carries one, sequence numbers are checked without the
 * exemption for 0), in zero-length packets (delivered), in pages that start with the tail of a lost packet (their packets are cut
 * from the wrong bytes, as the reference cuts them) and in continued packets (no granule position, no end-of-stream mark). */
int nvh_ogg_demux_forward(const uint8_t *bytes, size_t len, int stream_index, uint8_t *pkt_bytes, int64_t pkt_bytes_cap,
                          int64_t *offsets, int64_t *granules, uint8_t *flags, int pkt_cap, int *npackets,
                          int64_t *total_bytes, int *nstreams);

/* ---- seeking ----
 * Page table of one logical stream plus the reference's seek search over it:
 *   StreamPageReader.FindPage / FindPageBisection / FindPageForward   Ogg/StreamPageReader.cs:122-264
 *   PacketProvider.SeekTo / FindPacket / granule-bug workaround        Ogg/PacketProvider.cs:56-260
 *   PacketProvider.NormalizePacketIndex                                 Ogg/PacketProvider.cs:262-295
 * in the state the reference's reader is in once it has read every page of the stream.  The index copies what it needs;
 * `bytes` may be released after nvh_ogg_index_open returns. */
typedef struct nvh_ogg_index nvh_ogg_index;
int nvh_ogg_index_open(const uint8_t *bytes, size_t len, int stream_index, nvh_ogg_index **out);
void nvh_ogg_index_close(nvh_ogg_index *ix);
/* pages and packets of the stream, StreamPageReader.FirstDataPageIndex (-1: none), MaxGranulePosition, HasAllPages */
int nvh_ogg_index_info(const nvh_ogg_index *ix, int *npages, int *npackets, int *first_data_page, int64_t *max_granule,
                       int *has_all_pages);
/* StreamPageReader.GetPage (Ogg/StreamPageReader.cs:292-377): granule position, flags (1 resync, 2 continuation, 4 continued),
 * packet count, and the index (in nvh_ogg_demux_stream's packet list) of the first packet that starts on the page (-1: none) */
int nvh_ogg_index_page(const nvh_ogg_index *ix, int page, int64_t *granule, int *flags, int *packet_count, int *first_packet);
/* IPacketProvider.SeekTo(granulePos, preRoll, getPacketGranuleCount) (Ogg/PacketProvider.cs:56-72) with the stream's
 * StreamDecoder.GetPacketGranules (StreamDecoder.cs:630-647) as the callback: *packet_index = position in the packet list of the
 * packet GetNextPacket returns next, *granule_out = the method's return value.  NVH_ERR_ARGUMENT where the reference throws
 * ArgumentOutOfRangeException, NVH_ERR_INVALID_DATA for its InvalidDataExceptions, NVH_ERR_RUNTIME for an index fault. */
int nvh_ogg_seek(const nvh_ogg_index *ix, const nvh_stream *s, int64_t granule_pos, int pre_roll, int64_t *packet_index,
                 int64_t *granule_out);

/* ---- multi-GPU: the gather of the file-parallel corpus transcode (SURVEY 8 e) ----
 * StreamDecoder holds per-stream state only (StreamDecoder.cs:35-39), so a corpus shards by file: every process decodes its
 * files on its own GPU (one nvh_ctx per process) and the PCM is gathered on one of them -- the path's one collective, through
 * RCCL over xGMI.  Python callers have nvorbis_amd.corpus (torch.distributed); these entry points are the same exchange for a
 * host without it (the C# host: INTEGRATION.md, "Eight GPUs").  RCCL is loaded on first use (dlopen); NVH_ERR_UNSUPPORTED when
 * it is not installed, NVH_ERR_DEVICE for an RCCL failure (nvh_last_hip_error() = 10000 + its ncclResult_t).
 *   nvh_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other processes by whatever
 *                        means the host has (a file, a socket, its job launcher);
 *   nvh_comm_create      every rank, with the same id (ncclCommInitRank on the context's device; collective: returns when
 *                        all `world` ranks have called it);
 *   nvh_comm_allgather_i64  `n` 64-bit words per rank -- e.g. the number of floats of each file it decoded -- to every rank
 *                        (host arrays: mine[n], all[world * n], rank-major);
 *   nvh_comm_gather_pcm  rank r's `send_count` floats at d_send (HBM) arrive at d_recv + sum(counts[0..r-1]) on `root`
 *                        (d_recv: HBM, sum(counts) floats, ignored elsewhere); counts[world] = every rank's total, the same on every rank.  One
 *                        group of point-to-point transfers on the context's stream, waited for.  The root's own part is a
 *                        device copy; NVH_GATHER_SELF_P2P sends it through RCCL as well (a one-rank check of the
 *                        point-to-point path on a single GPU). */
#define NVH_COMM_ID_BYTES 128
#define NVH_GATHER_SELF_P2P 1
typedef struct nvh_comm nvh_comm;
int nvh_comm_unique_id(uint8_t *id);
int nvh_comm_create(nvh_ctx *ctx, const uint8_t *id, int rank, int world, nvh_comm **out);
void nvh_comm_destroy(nvh_comm *comm);
int nvh_comm_info(const nvh_comm *comm, int *rank, int *world);
int nvh_comm_allgather_i64(nvh_comm *comm, const int64_t *mine, int n, int64_t *all);
int nvh_comm_gather_pcm(nvh_comm *comm, const float *d_send, int64_t send_count, float *d_recv, const int64_t *counts,
                        int root, int flags);

#ifdef __cplusplus
}
#endif
#endif /* NVORBIS_HIP_H */
