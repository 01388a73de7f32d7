// host_parse.h -- audio-packet bit parser + stream geometry state machine (product host code).
//
// Splits NVorbis' per-packet work at the only place it can be split: everything that consumes bits
// (Mode.GetPacketInfo, IFloor.Unpack, the classification / entry decode inside IResidue.Decode) runs
// here on the host and is recorded as a frame descriptor; everything that touches float vectors
// (WriteVectors adds, inverse coupling, IFloor.Apply, IMdct.Reverse, windowing, OverlapBuffers,
// interleave + clip) runs on the GPU from those descriptors.  Bit parsing never depends on a float
// result, so the parser can run arbitrarily far ahead of synthesis.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

#include "host_setup.h"
#include "nvh_parse_format.h"
#include "nvh_format.h"

namespace nvh {

// The packet bytes of a GPU-parse batch (light mode): a growable byte pool whose storage the owner of the batch may place in
// page-locked host memory (nvh_api.hip does, for streams with a device context), so that the upload to the device reads it where
// the parser wrote it -- a second copy of every packet into a staging block was 2 ms of a 6.5 ms cycle at 32 768 packets.
// grow == nullptr: plain heap.  grow(owner, old, old_cap, new_cap) returns storage of new_cap bytes (the pool copies the old
// content over and calls grow(owner, old, old_cap, 0) to give the old block back).
struct PacketPool {
  uint8_t* base = nullptr;
  size_t size = 0, cap = 0;
  void* owner = nullptr;
  uint8_t* (*grow)(void* owner, uint8_t* old, size_t old_cap, size_t new_cap) = nullptr;
  PacketPool() = default;
  PacketPool(const PacketPool&) = delete;
  PacketPool& operator=(const PacketPool&) = delete;
  PacketPool(PacketPool&& o) noexcept { *this = std::move(o); }
  PacketPool& operator=(PacketPool&& o) noexcept {
    if (this != &o) {
      release();
      base = o.base; size = o.size; cap = o.cap; owner = o.owner; grow = o.grow;
      o.base = nullptr; o.size = o.cap = 0;
    }
    return *this;
  }
  ~PacketPool() { release(); }
  void release();
  uint8_t* append(size_t n);  // n more bytes (uninitialised); nullptr when out of memory
  void clear() { size = 0; }
  bool empty() const { return size == 0; }
};

struct FrameBatch {
  std::vector<NvhFrame> frames;
  std::vector<NvhChan> chans;
  std::vector<NvhResPass> passes;
  std::vector<NvhResOp> ops;
  // per op: bits 0-14 = the next op (frame-relative index) that adds to the same partition/channel in a later
  // stage (NVH_LINK_NONE: none), bit 15 = this op has such a predecessor.  Lets a kernel lane walk all stages
  // of one partition in order without a barrier per stage (kernels_spectrum.hip).
  std::vector<uint16_t> op_link;
  bool links_ok = true;         // false: some frame has too many ops for the 15-bit links
  std::vector<uint16_t> entries;
  std::vector<uint16_t> posts;
  std::vector<float> coeffs;
  // GPU-parse mode (StreamParser::set_light): the packets themselves, word aligned and zero padded, and where each
  // frame's packet lies (parallel to `frames`; pseudo-frames carry an empty reference)
  PacketPool pkt_pool;
  std::vector<NvhPacketRef> pkt_refs;
  int64_t pcm_samples = 0;      // per-channel samples the batch emits
  bool sequential_ola = false;  // some overlap region reaches into a tail: apply overlaps in order
  bool clipped_unknown = true;
  void clear() {
    frames.clear(); chans.clear(); passes.clear(); ops.clear(); op_link.clear(); entries.clear(); posts.clear(); coeffs.clear(); pkt_pool.clear(); pkt_refs.clear();
    links_ok = true;
    pcm_samples = 0;
    sequential_ola = false;
  }
};

// packet flags: NVH_PKT_EOS / NVH_PKT_RESYNC of the public C ABI header

// Mirrors the integer half of StreamDecoder.Read / ReadNextPacket / DecodeNextPacket
// (StreamDecoder.cs:320-530): which samples each packet emits, where the previous tail overlaps,
// the EOS trim, the drain after a failed packet, SamplePosition bookkeeping.
class StreamParser {
 public:
  explicit StreamParser(const Setup* s) : s_(s) {}

  // Parse one audio packet.  Appends at most one frame (and possibly extends the previous frame's
  // emission when the packet fails: "drain", StreamDecoder.cs:352-356).  Returns NVH_OK or an error
  // that corresponds to an exception escaping ReadSamples in the reference.
  int push_packet(const uint8_t* data, int len, int64_t granule, int flags, FrameBatch& out);
  // The provider ran out of packets (GetNextPacket()==null, StreamDecoder.cs:472-475).
  int push_end(FrameBatch& out);
  // Call after a batch was handed to synthesis: following frames refer to the carried tail.
  void begin_batch();

  // IResidue.Decode on its own (Residue0.cs:119-178, fine-grained ABI): the bit-consuming half of one call, starting at
  // bit `bit_offset` of the packet, recorded as a single-frame batch whose only pass is this residue.
  int parse_residue(int residue_idx, const uint8_t* data, int len, int bit_offset, int block_size, FrameBatch& out,
                    int* bits_consumed);

  // Light mode: only the packet type, mode number and window flags are read here (frame geometry, overlap and
  // position bookkeeping); floors and residues are left to kernels_parse.hip, which receives the packet bytes.
  void set_light(bool on) { light_ = on; }
  bool light() const { return light_; }
  // _hasPosition / _currentPosition (StreamDecoder.cs:35-39): read and set by a caller that starts a decoder in the
  // middle of a stream (what SeekTo leaves behind, :562-628; the chunked decode of nvorbis_amd/corpus.py)
  bool has_position() const { return has_position_; }
  void set_position_state(bool has, int64_t pos) { has_position_ = has; position_ = pos; }
  bool eos() const { return eos_found_; }
  int64_t position() const { return position_; }   // IStreamDecoder.SamplePosition after everything parsed was read
  int64_t emitted() const { return emitted_; }

 private:
  int parse_audio(BitReader& p, FrameBatch& out, int* decoded);
  int decode_floor(int floor_idx, BitReader& p, FrameBatch& out, NvhChan& ch, bool* energy);
  int decode_residue(int residue_idx, BitReader& p, int block_size, FrameBatch& out, NvhResPass& pass, uint32_t frame_op_begin);
  void drain(FrameBatch& out);

  const Setup* s_;
  bool light_ = false;
  std::vector<int> scratch_part_word_;      // decode_residue's rows, kept between packets
  std::vector<int32_t> scratch_last_op_;
  // StreamDecoder.cs:30-39 state, integer part
  bool has_prev_buf_ = false;   // _prevPacketBuf != null
  int prev_start_ = 0, prev_end_ = 0, prev_stop_ = 0;
  int prev_frame_ = -1;         // index of the previous decoded frame in the current batch, -2 = carried
  int prev_n_ = 0;              // block size of the previous decoded frame
  uint32_t prev_window_off_ = 0;
  std::vector<uint8_t> prev_exec_;  // per channel, of the previous decoded frame
  bool has_position_ = false, eos_found_ = false;
  int64_t position_ = 0;        // _currentPosition + bufferedSamples
  int64_t emitted_ = 0;         // total samples emitted since open (per channel)
};

}  // namespace nvh
