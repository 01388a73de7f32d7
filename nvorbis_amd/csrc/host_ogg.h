// host_ogg.h -- packet list produced by the minimal forward-only Ogg demux (host_ogg.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace nvh {

// One page of the logical stream as StreamPageReader.GetPage reports it (Ogg/StreamPageReader.cs:292-377), plus what the
// seek search needs of its packet slots.
struct OggPageInfo {
  int64_t granule = 0;       // raw header value, -1 on a page that completes no packet
  bool resync = false;       // lost page sync or sequence-number jump (Ogg/StreamPageReader.cs:77-86)
  bool continuation = false; // header flag "continues a packet"
  bool continued = false;    // last lacing value 255
  int packet_count = 0;      // slots (a continuation tail counts as slot 0)
  std::vector<int32_t> flat; // per slot: index of the packet of the demuxed list that STARTS there, or -1
  std::vector<uint8_t> head; // per slot: its first (up to) 8 bytes, 8 per slot
  std::vector<int32_t> len;  // per slot: byte length of the fragment on this page
};

struct OggPackets {
  std::vector<uint8_t> bytes;
  std::vector<int64_t> offs;      // n + 1 byte offsets
  std::vector<int64_t> granule;   // -1 = packet carries no granule position
  std::vector<uint8_t> flags;     // NVH_PKT_EOS | NVH_PKT_RESYNC
  // page table (filled when ogg_demux is asked for it)
  std::vector<OggPageInfo> pages;
  int first_data_page = -1;       // StreamPageReader._firstDataPageIndex: first page with a granule position > 0
  bool has_all_pages = false;     // the end-of-stream page was seen
  int64_t max_granule = 0;        // StreamPageReader._maxGranulePos
};

// The index form of ogg_demux: the same page walk and packet rules without the page checksums and without the packets' bodies --
// what a pass needs that only asks how many samples the stream decodes to (page headers + lacing values are ~1 % of a file).
// The first `full_first` packets (the Vorbis headers) are delivered whole, every other packet as its first `head_bytes` bytes
// (the packet type, mode number and window flags are in the first two); payload_bytes = the bytes the packets really have.
// A damaged page passes unnoticed here: whoever decodes the file demultiplexes it with the checksums and compares the packet
// count and payload_bytes (a refused page changes both).
struct OggIndexMode {
  int head_bytes = 8;
  int full_first = 3;
  int64_t payload_bytes = 0;  // out
};

// Packets of logical stream `stream_index` (0 = the first one whose page appears); *nstreams = how many there are.
int ogg_demux(const uint8_t* bytes, size_t len, OggPackets& out, int stream_index = 0, int* nstreams = nullptr, bool want_pages = false,
              OggIndexMode* index_mode = nullptr);

// The same for a source that cannot seek: ForwardOnlyPageReader + ForwardOnlyPacketProvider (Ogg/ForwardOnlyPageReader.cs,
// Ogg/ForwardOnlyPacketProvider.cs:36-67, 119-290); the differences are listed in host_ogg.cpp.  A granule position of -1 in
// `granule` stands for "none" as well as for a page value of -1.
int ogg_demux_forward(const uint8_t* bytes, size_t len, OggPackets& out, int stream_index = 0, int* nstreams = nullptr);

// PacketProvider.SeekTo (Ogg/PacketProvider.cs:56-72) over a demuxed stream with its page table, in the state the reference's
// reader is in once it has seen every page: the page search (StreamPageReader.FindPage, Ogg/StreamPageReader.cs:122-264), the
// packet search with the libvorbis granule workaround (Ogg/PacketProvider.cs:74-260) and NormalizePacketIndex (:262-295).
// granule_count(head bytes, length, is_resync) is StreamDecoder.GetPacketGranules (StreamDecoder.cs:630-647).
// *packet = index into the demuxed packet list of the packet GetNextPacket returns next; *granule_out = the returned position.
// NVH_ERR_ARGUMENT = ArgumentOutOfRangeException, NVH_ERR_INVALID_DATA = InvalidDataException, NVH_ERR_RUNTIME = an index
// fault of the managed code.
typedef int (*OggGranuleCount)(void* user, const uint8_t* head, int len, bool is_resync);
int ogg_seek(const OggPackets& ix, OggGranuleCount granule_count, void* user, int64_t granule_pos, int pre_roll, int64_t* packet,
             int64_t* granule_out);

}  // namespace nvh
