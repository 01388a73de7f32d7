// kernels_parse.hip -- the bit-consuming half of the decoder on the GPU: one lane per
// audio packet turns the packet's bits into the same frame descriptors host_parse.cpp produces (channel records,
// raw Floor1 posts, residue passes / ops / op links / VQ entry numbers), so that the synthesis kernels can follow
// without the packet ever being parsed on a host core.
//
// Reference behaviour followed (file:line under /root/reference/NVorbis/); the structure mirrors host_parse.cpp,
// which stays the definition (tests/test_gpu_parse.py compares the two descriptor streams field by field):
//   DataPacket.cs:150-283 (bit reader: zero-extended peeks, parked skips), Codebook.cs:294-320 (DecodeScalar),
//   Floor1.cs:135-184 (Unpack), Mapping.cs:95-134 (bit half of DecodePacket, ForceEnergy / ForceNoEnergy),
//   Residue0.cs:119-201, Residue1.cs:8-26, Residue2.cs:16-47 (classification + entry decode).
// What stays on the host in this mode is the per-stream integer state machine (Mode.GetPacketInfo, overlap
// geometry, sample positions): it needs the packet type, the mode number and the two window flags only.
//
// Huffman decoding is sequential inside a packet, so the parallelism is across packets.  Per-lane state is a 64-bit bit
// buffer refilled by aligned word loads (the host aligns and zero-pads every packet); the hot tables lie in LDS.  Output
// goes to fixed-capacity per-frame slabs.  The forms (nvh_launch.hip: batch_upload_gpu picks one per batch):
//   k_parse_slab_u       one packet per wavefront, wave-uniform control flow (batches of up to 4096 packets)
//   k_parse_slab_f       several packets per wavefront, one cursor per lane through the residue walk, ordinary packets only;
//   + k_parse_slab_c       the general body over the frames _f left (packets that end inside the residue, faults, other shapes)
//   + k_parse_slab_t       the rest of the slab (heads, entries, floors, header), one wavefront per packet
//   k_parse_slab / _g    several packets per wavefront in lockstep through the reference's loop nest (rounds 3-5; NVH_PARSE_CUR=0)
//   k_parse / _g         descriptors instead of slabs, for the stream shapes outside the slab kernels' contract
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels_common.h"
#include "nvh_parse_format.h"
#include "spectrum_dev.h"

namespace {

// status codes of include/nvorbis_hip.h used here
constexpr int kErrRuntime = -3;      // NVH_ERR_RUNTIME: the managed code would have thrown
constexpr int kErrUnsupported = -7;  // NVH_ERR_UNSUPPORTED

// Address spaces are kept apart by hand everywhere in this file: a pointer that may point to LDS *or* global memory
// compiles to FLAT loads, which cost a global-memory round trip even when they hit LDS and wait for every store in
// flight -- that alone made a symbol cost ~2500 cycles.
struct BitR {
  const uint32_t* w;   // packet words in global memory ...
  int lds_word;        // ... or (>= 0) their word offset inside the workgroup's LDS packet area
  uint32_t nwords, next;
  uint32_t total, pos;
  uint64_t buf;
  uint32_t avail;
  uint32_t ahead;      // word `next` of the packet, fetched when word next - 1 went into the buffer: a refill never waits for memory
  bool is_short;
};

template <bool LDS>
__device__ __forceinline__ uint32_t br_word(const BitR& b, const uint32_t* __restrict__ s_pkt, uint32_t i) {
  if (LDS) return i >= b.nwords ? 0u : s_pkt[b.lds_word + (int)i];  // compile-time choice: never a generic pointer
  // Global memory: the word is fetched whether or not the packet has it.  Index nwords -- the one word past the packet a reader
  // ever asks for -- is the next packet's first word or the pool's zero tail (nvh_launch.hip: eight zero bytes behind the pool), and
  // every use of a fetched word is behind `next < nwords`.  With the bounds test in front of the load the fetched value reaches
  // its register through a copy, and the copy waits for the load then and there: "fetched ahead" was a round trip to L2 in every
  // refill, ~700 cycles per symbol with several packets per wavefront (profiles/r06_cursor.txt).
  return b.w[i < b.nwords ? i : b.nwords];
}

template <bool LDS>
__device__ __forceinline__ void br_fill(BitR& b, const uint32_t* __restrict__ s_pkt) {
  while (b.avail <= 32 && b.next < b.nwords) {
    const uint32_t word = b.ahead;
    b.next++;
    b.ahead = br_word<LDS>(b, s_pkt, b.next);
    b.buf |= (uint64_t)word << b.avail;
    b.avail += 32;
  }
}

template <bool LDS>
__device__ __forceinline__ void br_init(BitR& b, const uint32_t* words, int lds_word, const uint32_t* __restrict__ s_pkt,
                                        uint32_t total_bits, uint32_t start) {
  b.w = words;
  b.lds_word = lds_word;
  b.total = total_bits;
  b.nwords = (total_bits + 31u) >> 5;
  b.pos = start < total_bits ? start : total_bits;
  b.is_short = false;
  b.next = b.pos >> 5;
  b.ahead = br_word<LDS>(b, s_pkt, b.next);
  b.buf = 0;
  b.avail = 0;
  br_fill<LDS>(b, s_pkt);
  const uint32_t s = b.pos & 31u;
  if (s) {
    b.buf >>= s;
    b.avail = b.avail >= s ? b.avail - s : 0;
    br_fill<LDS>(b, s_pkt);
  }
}

// DataPacket.TryPeekBits (:168-205): the remaining bits zero-extended, `got` = how many were really there
__device__ __forceinline__ uint32_t br_peek(const BitR& b, int count, int* got) {
  if (count <= 0) { *got = 0; return 0; }
  const uint32_t remaining = b.total - b.pos;
  const uint32_t n = (uint32_t)count < remaining ? (uint32_t)count : remaining;
  *got = (int)n;
  if (n == 0) return 0;
  return (uint32_t)(b.buf & ((~0ull) >> (64 - n)));
}

// DataPacket.SkipBits (:247-280): past the end parks the cursor there and raises IsShort
template <bool LDS>
__device__ __forceinline__ void br_skip(BitR& b, int count, const uint32_t* __restrict__ s_pkt) {
  if (count <= 0) return;
  if (b.total - b.pos >= (uint32_t)count) {
    b.buf >>= count;
    b.avail -= (uint32_t)count;
    b.pos += (uint32_t)count;
    br_fill<LDS>(b, s_pkt);
  } else {
    b.pos = b.total;
    b.is_short = true;
    b.buf = 0;
    b.avail = 0;
    b.next = b.nwords;
  }
}

// The cursor moved q bits on in one step (the caller knows the bits are there): the buffer from the packet's words again,
// three independent reads, in the state br_fill leaves (more than 32 bits while words remain, the next word fetched ahead).
template <bool LDS>
__device__ __forceinline__ void br_jump(BitR& b, const uint32_t* __restrict__ s_pkt, uint32_t q) {
  const uint32_t np = b.pos + q, w = np >> 5, sh = np & 31u;
  const uint32_t W0 = br_word<LDS>(b, s_pkt, w), W1 = br_word<LDS>(b, s_pkt, w + 1u), W2 = br_word<LDS>(b, s_pkt, w + 2u);
  b.pos = np;
  b.buf = (((uint64_t)W1 << 32) | W0) >> sh;
  b.next = w + 2u < b.nwords ? w + 2u : b.nwords;
  b.avail = 32u * b.next - np;
  b.ahead = W2;
}

template <bool LDS>
__device__ __forceinline__ uint32_t br_read(BitR& b, int count, const uint32_t* __restrict__ s_pkt) {
  if (count == 0) return 0;
  int got;
  const uint32_t v = br_peek(b, count, &got);
  br_skip<LDS>(b, count, s_pkt);
  return v;
}

// Prefix-table entry of a book that did not make it into the LDS image.  Not inlined on purpose: next to the LDS read
// of the same table the optimiser would fold both into one FLAT load through a selected pointer.
__device__ __attribute__((noinline)) uint32_t prefix_from_global(const uint32_t* __restrict__ prefix, uint32_t index) {
  return prefix[index];
}

// Codebook.DecodeScalar (Codebook.cs:294-320).  -1 = no symbol; -2 = the reference would fault (null list)
// UNI (parse_body): the compiler takes the result of any call (and of any FLAT load) for divergent, which would make the whole
// parse divergent again; a volatile load from the global address space cannot be folded into the LDS read either, and is
// uniform when its address is.  (Checked with opt -passes='print<uniformity>': only the slab's tail is divergent.)
template <bool LDS, bool UNI = false>
__device__ __forceinline__ int decode_scalar(const NvhDevParse& T, const uint32_t* __restrict__ s_prefix, const uint32_t* __restrict__ s_pkt,
                                             const NvhPBook bk, BitR& p) {
  // The common case first: at least 32 bits left in the packet, the prefix table in LDS, the code no longer than the prefix.
  // Then TryPeekBits returns all the bits asked for, SkipBits cannot run off the end (a code has at most 32 bits), the low
  // word of the buffer is valid (br_fill keeps more than 32 bits there while words remain), and a symbol is one ds_read
  // between a mask and a 64-bit shift -- the general form below spends most of its instructions on the end-of-packet rules.
  if (p.total - p.pos >= 32u && bk.has_tree && bk.lds_off != 0xFFFFFFFFu) {
    const uint32_t node = s_prefix[bk.lds_off + ((uint32_t)p.buf & ((1u << bk.prefix_bits) - 1u))];
    if (node & 0x80u) {
      const uint32_t len = node & 0x7Fu;
      p.buf >>= len;
      p.avail -= len;
      p.pos += len;
      br_fill<LDS>(p, s_pkt);
      return (int)(node >> 8);
    }
  }
  int got;
  uint32_t data = br_peek(p, bk.prefix_bits, &got);
  if (got == 0) return -1;
  if (!bk.has_tree) return -2;
  uint32_t node;
  if (bk.lds_off != 0xFFFFFFFFu) node = s_prefix[bk.lds_off + data];  // ds_read
  else if (UNI) node = *(const volatile __attribute__((address_space(1))) uint32_t*)(unsigned long long)(T.prefix + (bk.prefix_off + data));  // (global, said so: a FLAT load counts as divergent too)
  else node = prefix_from_global(T.prefix, bk.prefix_off + data);
  if (node & 0x80u) {
    br_skip<LDS>(p, (int)(node & 0x7Fu), s_pkt);
    return (int)(node >> 8);
  }
  data = br_peek(p, bk.max_bits, &got);
  if (!bk.has_overflow) return -2;
  // the longer codes: the reference scans its whole overflow list for the first match; a code can only match a
  // peek whose low prefix_bits select its slot, so scanning that slot's group (same relative order) finds the same node
  uint32_t cnt = node & 0x7Fu;
  if (cnt != 0x7Fu && bk.ovf_lds != 0xFFFFFFFFu) {  // the slot's group lies in LDS (8-byte nodes: bits, value << 8 | length)
    const uint32_t g = bk.ovf_lds + 2u * (node >> 8);
    for (uint32_t k = 0; k < cnt; ++k) {
      const uint32_t bits = s_prefix[g + 2u * k], vl = s_prefix[g + 2u * k + 1u], len = vl & 0xFFu;
      if (bits == (data & ((1u << len) - 1u))) {
        br_skip<LDS>(p, (int)len, s_pkt);
        return (int)(vl >> 8);
      }
    }
    return -1;
  }
  const NvhPOverflow* ov = T.overflow + bk.ovf_off;
  if (cnt == 0x7Fu) {
    cnt = bk.ovf_count;
  } else {
    ov += bk.ovf_count + (node >> 8);
  }
  for (uint32_t k = 0; k < cnt; ++k) {
    const NvhPOverflow o = ov[k];
    if (o.bits == (data & o.mask)) {
      br_skip<LDS>(p, (int)o.length, s_pkt);
      return (int)o.value;
    }
  }
  return -1;
}

// Floor1.Unpack (Floor1.cs:135-184).  Writes the raw posts of one channel; returns 0 or an error code.
template <bool LDS, bool UNI = false>
__device__ __forceinline__ int decode_floor1(const NvhDevParse& T, const uint32_t* __restrict__ s_prefix, const uint32_t* __restrict__ s_pkt,
                                             const NvhPBook* books,
                                             const NvhPFloor1& f, BitR& p, uint16_t* __restrict__ posts,
                                             int* post_count_out) {
  int post_count = 0;
  int first_big = NVH_MAX_POSTS + 1;  // first post whose raw value does not fit 16 bits (documented limit)
  if (br_read<LDS>(p, 1, s_pkt) == 1) {
    post_count = 2;
    const uint32_t y0 = br_read<LDS>(p, f.y_bits, s_pkt), y1 = br_read<LDS>(p, f.y_bits, s_pkt);
    posts[0] = (uint16_t)y0;
    posts[1] = (uint16_t)y1;
    for (int i = 0; i < f.partition_count; i++) {
      const int cls = f.partition_class[i];
      const int cdim = f.class_dims[cls];
      const int cbits = f.class_sub_bits[cls];
      const uint32_t csub = (1u << cbits) - 1u;
      uint32_t cval = 0;
      if (cbits > 0) {
        const int r = decode_scalar<LDS, UNI>(T, s_prefix, s_pkt, books[f.class_master[cls]], p);
        if (r == -2) return kErrRuntime;
        cval = (uint32_t)r;
        if (cval == 0xFFFFFFFFu) {
          post_count = 0;
          break;
        }
      }
      bool stop = false;
      for (int j = 0; j < cdim; j++) {
        const int book = f.sub_book[cls][cval & csub];
        cval >>= cbits;
        if (book >= 0) {
          if (post_count >= NVH_MAX_POSTS) return kErrRuntime;  // Posts = new int[64]
          const int r = decode_scalar<LDS, UNI>(T, s_prefix, s_pkt, books[book], p);
          if (r == -2) return kErrRuntime;
          if (r == -1) {
            post_count = 0;
            stop = true;
            break;
          }
          if (r > 0xFFFF && post_count < first_big) first_big = post_count;
          posts[post_count] = (uint16_t)r;
        } else if (post_count < NVH_MAX_POSTS) {
          posts[post_count] = 0;  // Posts[] is zero-initialised and never written for a null book
        }
        ++post_count;
      }
      if (stop) break;
    }
  }
  if (post_count > NVH_MAX_POSTS) return kErrRuntime;  // UnwrapPosts would index past finalY[64]
  if (first_big < post_count) return kErrUnsupported;
  *post_count_out = post_count;
  return 0;
}

}  // namespace

// Slab mode: one channel's Floor1 curve as the synthesis kernels read it -- Floor1.UnwrapPosts (Floor1.cs:224-297) and the walk
// over the sorted, flagged posts (:196-216) by the whole wavefront (lane = post; spectrum_dev.h: floor_prepare, the unwrap the
// descriptor kernels run per frame), then the segments with their signed 32.32 step per bin (kernels_synth.hip: floor_walk_fx;
// host_slab.cpp writes the same for the host parser) and the segment of every group of four bins.  Returns the segment count.
__device__ __forceinline__ int floor_to_slab_wave(FloorScratch* Q, const NvhDevFloor1* __restrict__ F, const uint16_t* __restrict__ posts,
                                                  int pc, int half, const uint32_t* __restrict__ recip, int* err_word,
                                                  uint4* __restrict__ out, int lane) {
  FloorLane L;
  L.mode = 1; L.pc = pc; L.levels = F->levels; L.level = 0; L.lo = 0; L.hi = 1; L.x = 0; L.x_lo = 0; L.x_hi = 1; L.val = 0;
  L.sorted = 0; L.x_sorted = 0; L.range = F->range; L.mult = F->multiplier; L.adx_magic = 0;
  if (lane < pc) {
    L.lo = F->l_neigh[lane]; L.hi = F->h_neigh[lane]; L.level = F->level[lane]; L.x = F->x_list[lane];
    L.val = posts[lane];
    L.sorted = F->sort_idx[lane]; L.x_lo = F->x_lo[lane]; L.x_hi = F->x_hi[lane]; L.x_sorted = F->x_sorted[lane];
    L.adx_magic = F->adx_magic[lane];
  }
  floor_prepare(Q, L, lane, half, err_word, recip);
  sp_wave_sync();
  const int ns = Q->nseg;
  for (int i = lane; i < ns; i += 64) {
    // (x, xend, y, b, |dy| mod adx, +-adx) -> (x, xend, y, signed 32.32 step per bin)
    const FloorSeg q = Q->seg[i];
    const int sadx = (int)q.ady_adx >> 16;
    const unsigned adx = (unsigned)(sadx < 0 ? -sadx : sadx), r = q.ady_adx & 0xFFFFu;
    const unsigned ab = (unsigned)(q.b < 0 ? -q.b : q.b);
    const unsigned long long fr32 = adx ? (((unsigned long long)r << 32) + adx - 1) / adx : 0ull;  // r < adx: below 2^32
    unsigned long long Fx = ((unsigned long long)ab << 32) + fr32;
    if (sadx < 0) Fx = 0ull - Fx;
    out[i] = make_uint4(q.x_xend, (unsigned)q.y, (unsigned)Fx, (unsigned)(Fx >> 32));
  }
  // segment index of every group of four bins: the last segment that starts at or before the group's first bin
  uint8_t* tab = reinterpret_cast<uint8_t*>(out + ns);
  const int ngroups = half >> 2;
  for (int gq = lane; gq < ((ngroups + 15) & ~15); gq += 64) {
    int sg = 0;
    if (gq < ngroups) {
      const int x0 = gq << 2;
      for (int step = 64; step > 0; step >>= 1) {
        const int cand = sg + step;
        if (cand < ns && (int)(Q->seg[cand].x_xend & 0xFFFFu) <= x0) sg = cand;
      }
    }
    tab[gq] = (uint8_t)sg;
  }
  sp_wave_sync();  // the next channel reuses the scratch block
  return ns;
}

// One lane per frame of the batch.  Frames with n == 0 (drain pseudo-frames) have no packet.
// Slab layout: frame f owns passes [f*cap_pass, +cap_pass), ops / op_link [f*cap_ops, +cap_ops), entries
// [f*cap_ent, +cap_ent), posts [(f*channels + c) * NVH_MAX_POSTS, +NVH_MAX_POSTS), and two int scratch rows of cap_parts.
#define NVH_PARSE_MAX_WAVES 16  // wavefronts per k_parse workgroup (they share the LDS tables): blockDim.x / 64
// LDS: the packets and the residue walk's scratch rows of this workgroup live in LDS (k_parse), else in global memory
// (k_parse_g: batches with a packet too long for that).  A compile-time switch, so that no pointer is ever generic.
// SLAB: the lane writes the synthesis kernels' slab of its frame itself (nvh_format.h: NvhSlabHdr; section order header | records
// | heads | entries | floors): every vector write goes straight to its chain-major record, and behind the parse -- the lanes of
// the wavefront together again -- the floors (floor_to_slab), the heads, the entries and the header follow.  `ops` then only
// and `op_link` are not used.  Without SLAB: the
// descriptors of rounds 1-3, for the stream shapes the descriptor kernels serve.
// UNI: one packet per wavefront (the launch shape of every batch of up to 4096 packets), and the compiler is told so: the
// packet index comes through readfirstlane, every value of the parse derives from it, and so all 64 lanes run the packet's
// decode side by side with the same values -- the loops' conditions are wave-uniform and compile to scalar branches, where the
// one-lane-of-64 form pays for every level of its divergent loop nest in exec-mask arithmetic (half of the instructions of the
// entry loop, more in the levels around it).  Stores of the parse are issued by all lanes with one address and one value;
// behind the parse lane 0 is the packet's lane, as before.
// CUR (slab mode, several packets per wavefront): the residue walk as one cursor per lane -- see the comment at the walk.
// PHASE: 0 = the whole job in one kernel; 1 = the parse alone, several packets per wavefront, its per-packet results handed over
// in `handover` (NVH_PHO_* words per frame); 2 = the rest of the slab (heads, entries, floors, header) from that hand-over, ONE
// packet per wavefront: the floors are written by a whole wavefront per channel (lane = post), which in one kernel would happen
// packet after packet for every lane of the parse's wavefront -- as long as the parse itself once its lanes run side by side.
#define NVH_PHO_WORDS 16
#define NVH_PHO_BAIL 0x7FFFFFF0u  // word 0 of a frame's hand-over: k_parse_slab_f left this packet to the general body (see there)
#define NVH_PSTG 32  // entries of one vector a lane of the cursor walk collects in LDS (a longer vector stores straight to memory)
template <bool LDS, bool SLAB, bool UNI = false, bool CUR = false, int PHASE = 0>
__device__ __forceinline__ void parse_body(const NvhDevParse& T, const uint8_t* __restrict__ pkt_pool, const NvhPacketRef* __restrict__ refs, int nframes,
        NvhFrame* __restrict__ frames, NvhChan* __restrict__ chans, NvhResPass* __restrict__ passes, NvhResOp* __restrict__ ops,
        uint16_t* __restrict__ op_link, uint16_t* __restrict__ entries, uint16_t* __restrict__ posts, int* __restrict__ scratch,
        NvhParseResult* __restrict__ result, int lanes_arg, int scratch_words, int pkt_words, uint4* __restrict__ slabs,
        const int* __restrict__ order, uint32_t* __restrict__ handover NVH_DBG_PARAMS) {
  static_assert(PHASE == 0 || (SLAB && !LDS && !UNI), "the split form: slab mode, rows and packets in global memory");
  static_assert(!CUR || (SLAB && !UNI && !LDS), "the cursor walk: slab mode, several packets per wavefront, rows and packets in global memory");
  constexpr bool WAVE1 = UNI || PHASE == 2;  // one packet per wavefront, the packet index uniform
  const int lanes = WAVE1 ? 1 : lanes_arg;
  const int tab_words = PHASE == 2 ? 0 : T.lds_words;  // (the tail reads no Huffman table)
  // PHASE 1 behind k_parse_slab_f (pkt_words < 0): only the frames that kernel marked; a workgroup without one leaves at once
  const bool redo_only = PHASE == 1 && pkt_words < 0;
  if constexpr (PHASE == 1) {
    if (redo_only) {
      const int w0 = (int)threadIdx.x >> 6, l0 = (int)threadIdx.x & 63;
      const int fi = (int)blockIdx.x * ((int)(blockDim.x >> 6) * lanes_arg) + w0 * lanes_arg + l0;
      bool marked = l0 < lanes_arg && fi < nframes;
      if (marked) marked = handover[(long long)(order ? order[fi] : fi) * NVH_PHO_WORDS] == NVH_PHO_BAIL;
      // (a vote through the first word of the dynamic LDS: __syncthreads_or would bring a static block along, and the kernel's
      // opt-in to a CU's whole LDS is sized for the dynamic one alone)
      extern __shared__ __attribute__((aligned(16))) uint32_t s_vote[];
      if (threadIdx.x == 0) s_vote[0] = 0u;
      __syncthreads();
      if (marked) s_vote[0] = 1u;
      __syncthreads();
      const bool any = s_vote[0] != 0u;
      __syncthreads();  // (the table image goes over this word next)
      if (!any) return;
    }
  }
#ifdef NVH_DEBUG
#define PM(bit) (!(phase_mask & ((bit) << 8)))  // profiling builds: NVH_DEBUG_SPECTRUM_MASK = 15 + 256 * (pieces to leave out)
#else
#define PM(bit) true
#endif
  // hot Huffman tables into LDS (every lane of the wavefront helps, then lanes without a frame leave)
  extern __shared__ __attribute__((aligned(16))) uint32_t s_prefix[];
  uint32_t* s_meta = s_prefix + tab_words;  // books | floors | residues | mappings, as in the arena
  for (int i = threadIdx.x; i < tab_words; i += (int)blockDim.x) s_prefix[i] = T.lds_image[i];
  {
    const uint32_t* gm = reinterpret_cast<const uint32_t*>(T.books);
    for (int i = threadIdx.x; i < T.meta_words; i += (int)blockDim.x) s_meta[i] = gm[i];
  }
  __syncthreads();
  const NvhPBook* books = reinterpret_cast<const NvhPBook*>(s_meta);
  const NvhPFloor1* floors = reinterpret_cast<const NvhPFloor1*>(reinterpret_cast<const uint8_t*>(s_meta) + T.meta_floors_off);
  const NvhPResidue* residues = reinterpret_cast<const NvhPResidue*>(reinterpret_cast<const uint8_t*>(s_meta) + T.meta_residues_off);
  const NvhPMapping* mappings = reinterpret_cast<const NvhPMapping*>(reinterpret_cast<const uint8_t*>(s_meta) + T.meta_mappings_off);
  auto ipool_at = [&](uint32_t i) { return (int)T.ipool[i]; };
  // per-lane LDS (when the host found room): the residue walk's two scratch rows, and the packet itself -- the bit
  // reader and the class words are on every symbol's dependency chain, and a global round trip costs ~10x an LDS one
  int* s_lane = reinterpret_cast<int*>(s_meta + T.meta_words);          // [packets per workgroup][scratch_words]
  // (CUR: no per-lane rows or packets in LDS; the area holds the entry staging of the cursor walk, NVH_PSTG entries per lane)
  uint16_t* const s_stage16 = reinterpret_cast<uint16_t*>(s_meta + T.meta_words);
  uint32_t* s_pkt = reinterpret_cast<uint32_t*>(s_lane + (int)(blockDim.x >> 6) * lanes * scratch_words);  // [packets per workgroup][pkt_words]
  // `lanes` packets per wavefront, blockDim.x / 64 wavefronts per workgroup.  Packets follow different paths through
  // this code, so the lanes of a wavefront run mostly one after the other, and a lone wavefront issues an instruction
  // every ~5 cycles at best: the host picks few packets per wavefront and ~2 wavefronts per SIMD for small batches
  // (a 4096-packet batch at 64 per wavefront would sit on 64 of 1024 SIMDs) and fills wavefronts up for large ones.
  const int wave = WAVE1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  // (CUR: the per-lane LDS rows of the cursor walk are indexed by packet of the workgroup, not by thread: with 32 packets per
  // wavefront twice as many wavefronts share a copy of the tables)
  const int nl = (int)(blockDim.x >> 6) * lanes, li = wave * lanes + lane;
  // UNI: the wavefront's packet (uniform)
  const int f_u = (int)blockIdx.x * (int)(blockDim.x >> 6) + wave;
  const bool valid_u = f_u < nframes;
  // slab mode keeps the lanes without a packet alive: behind the parse the whole wavefront works on the floors of its packets
  const bool active = WAVE1 ? (lane == 0 && valid_u) : (lane < lanes && blockIdx.x * ((int)(blockDim.x >> 6) * lanes) + wave * lanes + lane < nframes);
  const bool parses = WAVE1 ? valid_u : active;  // takes part in the parse of a packet
  if ((!SLAB || PHASE == 1) && !parses) return;
  const int slot = WAVE1 ? wave : wave * lanes + (active ? lane : 0);  // packet of this workgroup
  // Which frame that is: the host hands the frames over longest packet first (`order`), so that the packets of a wavefront are of
  // a size -- its lanes run side by side only while all of them have symbols left -- and the launch ends on short ones.
  const int f_idx = WAVE1 ? (valid_u ? f_u : 0) : (active ? blockIdx.x * ((int)(blockDim.x >> 6) * lanes) + slot : 0);
  const int f = (order && PHASE != 2) ? order[f_idx] : f_idx;
  if (redo_only && handover[(long long)f * NVH_PHO_WORDS] != NVH_PHO_BAIL) return;  // (no barrier below in this form)
#ifdef NVH_DEBUG
#define PT_T(k) do { if (dbg && active) dbg[(long long)f * 24 + (PHASE == 2 ? 12 : 0) + (k)] = clock64(); } while (0)
#define PT_ACC_BEGIN() const long long pt_t0 = clock64()
#define PT_ACC_END(k) pt_acc[k] += clock64() - pt_t0
  long long pt_acc[3] = {0, 0, 0};  // cycles inside: the entry loops of the vectors, the class words, the cursor walk's steps between vectors
  long long pt_rounds = 0;          // vectors decoded (cursor walk: rounds of its outer loop)
#else
#define PT_T(k) do { } while (0)
#define PT_ACC_BEGIN() do { } while (0)
#define PT_ACC_END(k) do { } while (0)
#endif
  PT_T(0);
  NvhFrame fr;
  if (parses) {
    fr = frames[f];
  } else {
    fr.n = 0; fr.mapping = 0; fr.mdct_slot = 0;
  }
  const int nch = T.channels;
  NvhChan* ch_out = chans + (long long)f * nch;
  const uint32_t op_base = (uint32_t)f * (uint32_t)T.cap_ops, ent_base = (uint32_t)f * (uint32_t)T.cap_ent;
  const uint32_t pass_base = (uint32_t)f * (uint32_t)T.cap_pass;
  int err = 0;
  uint32_t nops = 0, nent = 0, npass = 0;
  uint32_t exec_mask = 0;
  bool links_ok = true;
  // slab mode: this frame's slab, its record area (right behind the header), the chains allocated so far
  uint4* const slab = SLAB ? slabs + (long long)f * T.slab_stride_vecs : nullptr;
  uint2* const recs = SLAB ? reinterpret_cast<uint2*>(slab + NVH_SLAB_HDR_VECS) : nullptr;
  uint32_t nrec_alloc = 0, nheads = 0;
  unsigned long long pcs = 0;  // post count of every channel, 7 bits each (at most eight channels)
  int s_rtype = 0, s_rch = 1, s_psz = 0, s_rbegin = 0, s_npass = 0, s_parts = 0, s_chs = 1, s_b1 = 0, s_res = 0;
  // streams of the general bin walk (T.slab_general): one chain-start row per residue pass, two words in front of each -- the
  // pass's residue and its partition count -- from which the tail below builds the heads and the group list
  const bool gen_rows = SLAB && T.slab_general != 0;
  const int row_stride = T.cap_parts + 2;
  int* const g_rows_base = scratch + (long long)f * T.row_words;
  int* const l_rows_base = reinterpret_cast<int*>(s_meta + T.meta_words) + slot * scratch_words;

  // hand-over between the two kernels of the split form (PHASE 1 -> 2), NVH_PHO_WORDS words per frame:
  //   0 err | 1 nrec_alloc | 2 nent | 3 nops | 4 npass | 5 exec_mask | 6, 7 pcs | 8 s_rtype | 9 s_rch | 10 s_psz | 11 s_rbegin |
  //   12 s_npass | 13 s_parts | 14 s_chs | s_b1 << 8 | links_ok << 16 | 15 s_res
  uint32_t* const ho = PHASE != 0 ? handover + (long long)f * NVH_PHO_WORDS : nullptr;
  if constexpr (PHASE == 2) {
    if (parses) {  // (uniform: every lane of the wavefront holds the packet's state, lane 0 is the packet's lane below)
      const uint4 h0 = *reinterpret_cast<const uint4*>(ho), h1 = *reinterpret_cast<const uint4*>(ho + 4),
                  h2 = *reinterpret_cast<const uint4*>(ho + 8), h3 = *reinterpret_cast<const uint4*>(ho + 12);
      err = (int)h0.x; nrec_alloc = h0.y; nent = h0.z; nops = h0.w;
      npass = h1.x; exec_mask = h1.y; pcs = ((unsigned long long)h1.w << 32) | h1.z;
      s_rtype = (int)h2.x; s_rch = (int)h2.y; s_psz = (int)h2.z; s_rbegin = (int)h2.w;
      s_npass = (int)h3.x; s_parts = (int)h3.y; s_chs = (int)(h3.z & 0xFFu); s_b1 = (int)((h3.z >> 8) & 0xFFu);
      links_ok = ((h3.z >> 16) & 1u) != 0; s_res = (int)h3.w;
    }
  }
  if (PHASE != 2 && fr.n != 0) {
    const NvhPacketRef ref = refs[f];
    BitR p;
    const uint32_t* pw = reinterpret_cast<const uint32_t*>(pkt_pool + ref.byte_off);
    const int pkt_nwords = (int)((ref.bit_len + 31u) >> 5);
    int lds_word = -1;
    if (LDS) {
      lds_word = slot * pkt_words;
      for (int i = 0; i < pkt_nwords; i++) s_pkt[lds_word + i] = pw[i];  // independent loads: one latency for the lot
    }
    br_init<LDS>(p, pw, lds_word, s_pkt, ref.bit_len, ref.bit_pos);
    const NvhPMapping& map = mappings[fr.mapping];
    PT_T(1);

    // ---- floors (Mapping.cs:95-111) ----
    uint32_t energy = 0;
    for (int c = 0; c < nch && !err; c++) {
      const int fl = map.chan_floor[c];
      int pc = 0;
      uint16_t* my_posts = posts + ((long long)f * nch + c) * NVH_MAX_POSTS;
      err = decode_floor1<LDS, UNI>(T, s_prefix, s_pkt, books, floors[fl], p, my_posts, &pc);
      NvhChan cn;
      cn.exec = 0;
      cn.floor = (uint8_t)fl;
      cn.post_count = (uint8_t)pc;
      cn.ov_exec = 0;
      cn.data_off = (uint32_t)(((long long)f * nch + c) * NVH_MAX_POSTS);
      cn.amp = 0.0f;
      ch_out[c] = cn;
      if (pc > 0) energy |= 1u << c;
      pcs |= (unsigned long long)(pc & 0x7F) << (7 * c);
    }
    const bool any_execute = energy != 0;  // computed before ForceEnergy (quirk B-5)
    uint32_t force_e = 0, force_no = 0;
    if (!err) {
      for (int i = 0; i < map.coupling_steps; i++) {  // Mapping.cs:112-119
        const uint32_t a = 1u << map.coupling_ang[i], m = 1u << map.coupling_mag[i];
        if (((force_e | energy) & ~force_no) & (a | m)) force_e |= a | m;
      }
    }
    PT_T(2);
    // ---- residues (Mapping.cs:122-134; Residue0.Decode :119-178) ----
    for (int sm = 0; sm < map.submaps && !err; sm++) {
      for (int j = 0; j < nch; j++)
        if (map.submap_floor[sm] != map.chan_floor[j] || map.submap_residue[sm] != map.chan_residue[j]) force_no |= 1u << j;
      if (!any_execute) continue;
      const int residue_idx = map.submap_residue[sm];
      const NvhPResidue& r = residues[residue_idx];
      // (the scalars of the record by value: `r` lies in LDS, which the walk also writes, so every use would be a reload)
      const int r_type = r.type, r_begin = r.begin, r_psize = r.partition_size, r_chs = r.channels, r_rch = r.real_channels,
                r_stages = r.max_stages, r_partvals = r.partvals;
      const uint32_t r_rchm = r.rch_magic;
      const NvhPBook class_book = books[r.class_book];  // by value: registers, not an LDS reload per symbol
      const uint32_t dm_lds = r.decode_map_lds;         // the class decode map's copy in LDS (nvh_setup.hip), if it has one
      const uint32_t vis_lds = r.vis_lds;               // ... the visit descriptors'
      // The entries of one vector, one packet per wavefront (UNI): the 64 lanes are idle copies of one another, and a symbol of the
      // scalar loop is a chain of a dozen and more instructions on the one scalar unit 16 wavefronts share.  Here lane i looks up the
      // code that WOULD start at bit i of a 96-bit window (buffer + the word fetched ahead) -- one alignbit, one table read, for up
      // to 63 start positions at once -- and the scalar side only walks the chain of lengths: readlane, add, per symbol.  The lanes
      // on the chain then store their entries side by side.  Same symbols, same order, same stop rules as the scalar loop (a start
      // position needs 32 bits left in the packet, a code that resolves in the prefix table, a slot); whatever ends a window early
      // -- its width, a long code -- is taken up by the next window or by the long-code scan.  Returns true when it cannot go on
      // for a reason the caller's loops do not know (never, by the bit reader's invariants).
      auto decode_windows = [&](const uint32_t toff, const uint32_t pmask, uint16_t* __restrict__ eout, const int slots, int& done) -> bool {
        bool stuck = false;
        for (;;) {
          const uint32_t rem = p.total - p.pos;
          if (done >= slots || rem < 32u) break;
          const uint32_t av = p.avail;
          uint64_t lo = p.buf;
          uint32_t hi = p.ahead;
          if (av < 64u) {
            lo |= (uint64_t)p.ahead << av;
            hi = av > 32u ? p.ahead >> (64u - av) : 0u;
          }
          const uint32_t wbits = av + (p.next < p.nwords ? 32u : 0u);
          const uint32_t mbits = wbits < rem ? wbits : rem;
          if (mbits < 32u) { stuck = true; break; }  // (cannot happen: the buffer holds the packet's last bits, or more than 32)
          const uint32_t limit = mbits - 32u < 62u ? mbits - 32u : 62u;  // last start position of this window (lane 63 ends every chain)
          const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32);
          const uint32_t fa = lane < 32 ? w0 : w1, fb = lane < 32 ? w1 : hi;
          const uint32_t field = __builtin_amdgcn_alignbit(fb, fa, (uint32_t)lane & 31u);
          const uint32_t nodev = s_prefix[toff + (field & pmask)];
          const uint32_t lenv = ((nodev & 0x80u) != 0u && (uint32_t)lane <= limit) ? (nodev & 0x7Fu) : 0u;
          // (lanes behind `limit` hold 0, lane 63 among them: a position past the window reads a 0 and ends the chain)
          uint32_t q, cnt, t_lane, t_len;
          unsigned long long chain;
          const uint32_t want = (uint32_t)__builtin_amdgcn_readfirstlane(slots - done);  // >= 1 (uniform: said so for the asm's scalar operand)
          // do { len = lenv[min(q, 63)]; if (!len) break; chain |= 1 << q; q += len; } while (++cnt < want);
          // by hand: the compiler turns the two exits into a dozen condition-mask instructions per symbol, and this
          // loop IS the parse's cost (nine scalar-side instructions per symbol; no software wait states are due: the
          // lane select is written by the scalar unit, v_readlane's result is read by it)
          asm volatile(
              "s_mov_b32 %0, 0\n\t"
              "s_mov_b32 %1, 0\n\t"
              "s_mov_b64 %2, 0\n"
              "1:\n\t"
              "s_min_u32 %3, %0, 63\n\t"
              "v_readlane_b32 %4, %5, %3\n\t"
              "s_cmp_eq_u32 %4, 0\n\t"
              "s_cbranch_scc1 2f\n\t"
              "s_bitset1_b64 %2, %0\n\t"
              "s_add_u32 %0, %0, %4\n\t"
              "s_add_u32 %1, %1, 1\n\t"
              "s_cmp_lt_u32 %1, %6\n\t"
              "s_cbranch_scc1 1b\n"
              "2:\n\t"
              : "=&s"(q), "=&s"(cnt), "=&s"(chain), "=&s"(t_lane), "=&s"(t_len)
              : "v"(lenv), "s"(want)
              : "scc");
          if (cnt == 0u) break;
          if ((chain >> lane) & 1ull) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(chain >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)chain, 0u));
            eout[(uint32_t)done + rank] = (uint16_t)(nodev >> 8);
          }
          done += (int)cnt;
          br_jump<LDS>(p, s_pkt, q);
        }
        return stuck;
      };
      NvhResPass pass;
      pass.residue = residue_idx;
      for (int s = 0; s <= NVH_MAX_STAGES; s++) pass.op_begin[s] = op_base + nops;
      int block_size = fr.n;
      if (r_type == 2) block_size *= r_rch;  // Residue2.cs:16-21
      const int end = r.end < block_size / 2 ? r.end : block_size / 2;
      const int n = end - r_begin;
      bool ran = false;
      int stage = 0;
      if (n > 0) {
        ran = true;
        const int partition_count = n / r_psize;
        const int cdim = r.class_dims;
        if (cdim == 0) {
          err = kErrRuntime;
          break;
        }
        // the two scratch rows of the walk: class per [channel][partition] (a class word is expanded through the residue's decode map
        // when it is decoded: a visit then is one lookup away from its class, not two), last op per [partition][channel]
        int* g_rows = g_rows_base;
        int* l_rows = l_rows_base;
        auto row_get = [&](int i) { return LDS ? l_rows[i] : g_rows[i]; };
        auto row_set = [&](int i, int v) { if (LDS) l_rows[i] = v; else g_rows[i] = v; };
        const int last_base = gen_rows ? T.cap_parts + s_npass * row_stride + 2 : T.cap_parts;  // last_op row behind the part_word row
        const int pw_stride = partition_count > 0 ? partition_count : 1;
        for (int i = 0; i < r_chs * pw_stride; i++) row_set(i, -1);
        auto class_of = [&](int word, int d) {
          if constexpr (UNI) return (int)s_prefix[dm_lds + (uint32_t)(word * cdim + d)];
          else return dm_lds != 0xFFFFFFFFu ? (int)s_prefix[dm_lds + (uint32_t)(word * cdim + d)] : ipool_at(r.decode_map_off + (uint32_t)(word * cdim + d));
        };
        for (int i = 0; i < r_chs * (partition_count > 0 ? partition_count : 1); i++) row_set(last_base + i, -1);
        const int buflen = T.block1;  // float[ch][block1Size] (StreamDecoder.cs:498-505)
        bool stop = false;
        int stop_p = 0, stop_c = 0;  // slab mode: where the packet ran out (partition, channel), and whether that vector write was kept
        bool stop_pushed = false;
        if constexpr (CUR) {
          // ---- the walk as one cursor per lane (several packets per wavefront) ----
          // The loop nest below visits (stage, partition, channel) in lockstep over the lanes of a wavefront: at a visit only the
          // lanes whose partition class has a book in that stage decode -- on real material one in four to one in eight -- and the
          // others wait; the entry loops of a wavefront's packets ran, in effect, one after the other (64 packets per wavefront:
          // 18 times one packet's time).  Here a lane carries its own position (stage, partition_idx, c) through the same visit
          // order: it steps over the visits that decode nothing, stops at the next one that does, and then ALL lanes run their
          // entry loops together, each on its own vector.  Same reads, same order, same stop rules per packet (the code of a
          // visit is the nest's, statement for statement); only which lanes keep each other company changes.
          // What a step needs to know about a position -- its class, and where its chain of records starts -- comes from LDS and
          // from a running count: the class of every (channel, partition) is kept as a byte per lane (lane-strided words behind the
          // entry staging; the host checked the room), and since every stage walks the positions in the order stage 0 allocated
          // their chains in, a chain's first record is the sum of the chain lengths in front of it.  The rows in global memory
          // are still written (the tail kernel builds the heads from them, the end-of-packet fix-up below reads them), never read
          // here: a step that waits for a global load behind its own stores cost 2-5 k cycles, 180 steps per packet.
          uint8_t* const s_cls = reinterpret_cast<uint8_t*>(s_stage16) + (size_t)nl * (NVH_PSTG * 2);
          auto cls_at = [&](const int i) -> uint8_t& { return s_cls[4 * ((i >> 2) * nl + li) + (i & 3)]; };
          const uint32_t pass_rec0 = nrec_alloc;
          uint32_t run = pass_rec0, start = 0;
          int partition_idx = 0, c = 0, dimension_idx = 0;
          bool running = r_stages > 0;
          if (running) pass.op_begin[0] = op_base + nops;
          while (running) {
            int cls = 0, book_idx = -1;
            bool have = false;
#ifdef NVH_DEBUG
            const long long pt_adv0 = clock64();
            ++pt_rounds;
#endif
            while (!have) {
              if (c >= r_chs) {
                c = 0;
                ++partition_idx;
                if (++dimension_idx >= cdim) dimension_idx = 0;
              }
              if (partition_idx >= partition_count) {  // next cascade stage (Residue0.cs:132)
                ++stage;
                if (stage >= r_stages) break;
                pass.op_begin[stage] = op_base + nops;
                partition_idx = 0; c = 0; dimension_idx = 0;
                run = pass_rec0;
                continue;
              }
              if (stage == 0 && c == 0 && dimension_idx == 0) {  // the class words of this group of partitions (:137-150)
                PT_ACC_BEGIN();
                for (int cc = 0; cc < r_chs; cc++) {
                  const int idx = decode_scalar<LDS, false>(T, s_prefix, s_pkt, class_book, p);
                  if (idx == -2) {
                    err = kErrRuntime;
                    break;
                  }
                  if (idx >= 0 && idx < r_partvals) {
                    for (int d = 0; d < cdim && partition_idx + d < partition_count; d++) {
                      const int k = class_of(idx, d);
                      row_set(cc * pw_stride + partition_idx + d, k);
                      cls_at(cc * pw_stride + partition_idx + d) = (uint8_t)k;  // (class numbers are below NVH_MAX_CLASSES)
                    }
                  } else {
                    stop = true;
                    stop_p = partition_idx; stop_c = 0; stop_pushed = false;
                    break;
                  }
                }
                PT_ACC_END(1);
                if (stop || err) break;
              }
              // (a position is reached only behind its group's class words: the nest's "class not set" fault cannot happen)
              const int cl = (int)cls_at(c * pw_stride + partition_idx);
              const unsigned cm = r.book_mask[cl];  // the stages in which this class has a book (nvh_setup.hip: cascade bit and book)
              start = run;
              run += (uint32_t)__popc(cm);
              if (stage == 0) {  // the chain of this partition / channel (see the nest)
                row_set(last_base + partition_idx * r_chs + c, cm ? (int)start : -1);
                nrec_alloc = run;
                if (nrec_alloc > (uint32_t)T.cap_ops) {
                  err = kErrRuntime;
                  break;
                }
              }
              if (((cm >> stage) & 1u) == 0) {
                ++c;
                continue;
              }
              cls = cl;
              book_idx = r.books[cl][stage];
              have = true;
            }
#ifdef NVH_DEBUG
            pt_acc[2] += clock64() - pt_adv0;
#endif
            if (!have) {  // the last stage is through, the packet ran out in a class word, or the reference would have thrown
              running = false;
              break;
            }
            // ---- one visit: the entries of a vector (Residue0.cs:157-170) ----
            const int offset = r_begin + partition_idx * r_psize;
            const NvhPBook book = books[book_idx];
            const int dims = book.dims;
            if (dims == 0) {
              err = kErrRuntime;
              break;
            }
            if (partition_idx > 0xFFFF) {
              err = kErrUnsupported;
              break;
            }
            const uint32_t ent_off = ent_base + nent;
            bool push = false, bad = false;
            if (r_type == 0) {
              // Residue0.WriteVectors (:180-201): decode all entries first, add only if all decoded
              const int steps = r_psize / dims;
              const uint32_t mark = nent;
              if (nent + (uint32_t)steps > (uint32_t)T.cap_ent) {
                err = kErrRuntime;
                break;
              }
              for (int i = 0; i < steps; i++) {
                const int e = decode_scalar<LDS, false>(T, s_prefix, s_pkt, book, p);
                if (e == -2) {
                  err = kErrRuntime;
                  break;
                }
                if (e == -1) {
                  bad = true;
                  break;
                }
                entries[ent_base + nent++] = (uint16_t)e;
              }
              if (err) break;
              if (bad) {
                nent = mark;
                stop = true;
                stop_p = partition_idx; stop_c = c; stop_pushed = false;
                break;
              }
              if (offset + steps * dims > buflen) {
                err = kErrRuntime;
                break;
              }
              push = true;
            } else {
              // Residue1.WriteVectors (Residue1.cs:8-26) / Residue2.WriteVectors (Residue2.cs:23-47): vectors are added as they
              // are decoded; a failed decode keeps what was added so far
              const int slots = dims > 1 ? (int)__umulhi((uint32_t)(r_psize + dims - 1), book.dim_magic) : r_psize;
              if (nent + (uint32_t)slots > (uint32_t)T.cap_ent) {
                err = kErrRuntime;
                break;
              }
              int done = 0;
              PT_ACC_BEGIN();
              // fast form (see the nest): a symbol is a masked ds_read and a 64-bit shift while the packet has 32 bits left, the code
              // resolves in the book's LDS prefix table and slots remain; long codes from their slot's group.  The entries of the
              // vector are collected in LDS (NVH_PSTG per lane, lane-strided words) and leave behind the loop: a store per symbol
              // is a store in flight at every refill of the bit buffer, and the wait for the word fetched ahead is a wait for
              // every memory operation before it -- a round trip to L2 per symbol where the loop's own chain is one LDS read.
              if (book.has_tree && book.lds_off != 0xFFFFFFFFu) {
                const uint32_t pmask = (1u << book.prefix_bits) - 1u, toff = book.lds_off;
                uint16_t* __restrict__ eout = entries + ent_base + nent;
                uint32_t node = s_prefix[toff + ((uint32_t)p.buf & pmask)];
                auto consume = [&](const uint32_t len) {
                  p.buf >>= len;
                  p.avail -= len;
                  p.pos += len;
                  if (p.avail <= 32u && p.next < p.nwords) {  // (one word restores br_fill's invariant: len <= 32)
                    const uint32_t word = p.ahead;
                    p.next++;
                    p.ahead = br_word<LDS>(p, s_pkt, p.next);
                    p.buf |= (uint64_t)word << p.avail;
                    p.avail += 32u;
                  }
                };
                auto run = [&](auto staged) {
                  constexpr bool STG = decltype(staged)::value;
                  auto put = [&](const int i, const uint32_t v) {
                    if constexpr (STG) s_stage16[2 * ((i >> 1) * nl + li) + (i & 1)] = (uint16_t)v;
                    else eout[i] = (uint16_t)v;
                  };
                  for (;;) {
                    while ((bool)((int)(done < slots) & (int)(p.total - p.pos >= 32u) & (int)((node >> 7) & 1u))) {
                      consume(node & 0x7Fu);
                      put(done, node >> 8);
                      ++done;
                      node = s_prefix[toff + ((uint32_t)p.buf & pmask)];
                    }
                    if (!((int)(done < slots) & (int)(p.total - p.pos >= 32u) & (int)(book.has_overflow != 0))) break;
                    const uint32_t data = (uint32_t)p.buf & (book.max_bits >= 32 ? 0xFFFFFFFFu : (1u << book.max_bits) - 1u);
                    uint32_t cnt = node & 0x7Fu;
                    uint32_t hit_len = 0, hit_val = 0;
                    if (cnt != 0x7Fu && book.ovf_lds != 0xFFFFFFFFu) {
                      const uint32_t g = book.ovf_lds + 2u * (node >> 8);
                      for (uint32_t k = 0; k < cnt; ++k) {
                        const uint32_t bits = s_prefix[g + 2u * k], vl = s_prefix[g + 2u * k + 1u], len = vl & 0xFFu;
                        if (bits == (data & ((1u << len) - 1u))) {
                          hit_val = vl >> 8;
                          hit_len = len;
                          break;
                        }
                      }
                    } else {
                      const NvhPOverflow* __restrict__ ov = T.overflow + book.ovf_off;
                      if (cnt == 0x7Fu) cnt = book.ovf_count;
                      else ov += book.ovf_count + (node >> 8);
                      for (uint32_t k = 0; k < cnt; ++k) {
                        const uint4 o = *reinterpret_cast<const uint4*>(ov + k);  // bits, mask, value, length
                        if (o.x == (data & o.y)) {
                          hit_val = o.z;
                          hit_len = o.w;
                          break;
                        }
                      }
                    }
                    if (hit_len == 0u || hit_len > 32u) break;  // no match (or nothing this loop may skip): the general loop decides
                    consume(hit_len);
                    put(done, hit_val);
                    ++done;
                    node = s_prefix[toff + ((uint32_t)p.buf & pmask)];
                  }
                  if constexpr (STG) {
                    // the collected entries to their place: a lone leading one where the vector starts on an odd entry, then pairs
                    auto get = [&](const int i) { return (uint32_t)s_stage16[2 * ((i >> 1) * nl + li) + (i & 1)]; };
                    int k = 0;
                    if ((nent & 1u) && done > 0) {
                      eout[0] = (uint16_t)get(0);
                      k = 1;
                    }
                    for (; k + 1 < done; k += 2) *reinterpret_cast<uint32_t*>(eout + k) = get(k) | (get(k + 1) << 16);
                    if (k < done) eout[k] = (uint16_t)get(k);
                  }
                };
                if (slots <= NVH_PSTG) run(std::true_type{});
                else run(std::false_type{});
                nent += (uint32_t)done;
              }
              for (int i = done * dims; i < r_psize; i += dims) {
                const int e = decode_scalar<LDS, false>(T, s_prefix, s_pkt, book, p);
                if (e == -2) {
                  err = kErrRuntime;
                  break;
                }
                if (e == -1) {
                  bad = true;
                  break;
                }
                entries[ent_base + nent++] = (uint16_t)e;
                ++done;
              }
              PT_ACC_END(0);
              if (err) break;
              if (done > 0) {  // bounds of the adds the reference performed
                const int last = done * dims - 1;
                if (r_type == 1) {
                  if (offset + last >= buflen) err = kErrRuntime;
                } else {
                  const int ob = r_rch > 1 ? (int)__umulhi((uint32_t)offset, r_rchm) : offset;
                  const int lb = r_rch > 1 ? (int)__umulhi((uint32_t)last, r_rchm) : last;
                  if (ob + lb >= buflen) err = kErrRuntime;
                }
                if (err) break;
              }
              for (int i = done; i < slots; i++) entries[ent_base + nent++] = (uint16_t)NVH_ENTRY_SKIP;
              push = true;
            }
            if (push) {
              if (nops >= (uint32_t)T.cap_ops) {
                err = kErrRuntime;
                break;
              }
              const unsigned cm = r.book_mask[cls];
              const unsigned rank = (unsigned)__popc(cm & ((1u << stage) - 1u));
              const uint32_t rw[2] = {NVH_SLAB_REC(ent_off - ent_base, book.slab_dm16, book.slab_lat & 0xFFFFu, book.slab_lat >> 16, dims, c,
                                                   stage, (cm >> (stage + 1)) != 0)};
              if (PM(1)) recs[start + rank] = make_uint2(rw[0], rw[1]);
              ++nops;
            }
            if (bad) {
              stop = true;
              stop_p = partition_idx; stop_c = c; stop_pushed = true;
              break;
            }
            ++c;
          }
          if (stop || err) ++stage;  // (the nest leaves `stage` one past the stage the packet ended in)
        } else {
        for (; stage < r_stages && !stop && !err; stage++) {
          pass.op_begin[stage] = op_base + nops;
          for (int partition_idx = 0; partition_idx < partition_count && !stop && !err;) {
            if (stage == 0) {
              PT_ACC_BEGIN();
              for (int c = 0; c < r_chs; c++) {
                const int idx = decode_scalar<LDS, UNI>(T, s_prefix, s_pkt, class_book, p);
                if (idx == -2) {
                  err = kErrRuntime;
                  break;
                }
                if (idx >= 0 && idx < r_partvals) {
                  for (int d = 0; d < cdim && partition_idx + d < partition_count; d++) row_set(c * pw_stride + partition_idx + d, class_of(idx, d));
                } else {
                  stop = true;
                  stop_p = partition_idx; stop_c = 0; stop_pushed = false;
                  break;
                }
              }
              PT_ACC_END(1);
              if (stop || err) break;
            }
            for (int dimension_idx = 0; partition_idx < partition_count && dimension_idx < cdim && !stop && !err;
                 dimension_idx++, partition_idx++) {
              const int offset = r_begin + partition_idx * r_psize;
              for (int c = 0; c < r_chs; c++) {
                // (UNI: launched only for setups whose decode maps are all in LDS: no global load on this path)
                const int cls = row_get(c * pw_stride + partition_idx);
                if (cls < 0) {
                  err = kErrRuntime;  // NullReferenceException on partWordCache
                  break;
                }
                if (SLAB && stage == 0) {
                  // the chain of this partition / channel: one record per cascade stage that has a book, consecutive, allocated
                  // now that its class is known (stage 0 visits every partition in order)
                  // (only the allocation happens here, where the lanes of the wavefront run one after the other; the list of
                  // chain heads is written behind the parse, from these rows)
                  const unsigned cm = r.book_mask[cls];
                  row_set(last_base + partition_idx * r_chs + c, cm ? (int)nrec_alloc : -1);
                  nrec_alloc += (uint32_t)__popc(cm);
                  if (nrec_alloc > (uint32_t)T.cap_ops) {
                    err = kErrRuntime;
                    break;
                  }
                }
                int fast_done = 0;
                if constexpr (UNI && SLAB && LDS) {
                  // The visit from its descriptor (nvh_parse_format.h: NVH_PVIS_*; the launch guarantees the table): one 16-byte read
                  // instead of the cascade test, the book number, the book's record and the slot arithmetic.  A vector that decodes
                  // completely in windows -- all of them, on real material -- is done here: its record, the counters, on to the next
                  // visit.  Whatever is left of one that does not (a long code, the packet's last bits) the general code below
                  // finishes, from the entries decoded so far.
                  const uint4 V = *reinterpret_cast<const uint4*>(s_prefix + vis_lds + 4u * (uint32_t)(cls * NVH_MAX_STAGES + stage));
                  if (V.x == NVH_PVIS_NONE) continue;
                  const uint32_t vslots = V.y & 0xFFFFu;
                  if (V.x != NVH_PVIS_SLOW && nent + vslots <= (uint32_t)T.cap_ent && nops < (uint32_t)T.cap_ops && partition_idx <= 0xFFFF) {
                    PT_ACC_BEGIN();
                    int wdone = 0;
                    const bool stuck = decode_windows(V.x & 0xFFFFFFu, (1u << (V.x >> 24)) - 1u, entries + ent_base + nent, (int)vslots, wdone);
                    PT_ACC_END(0);
                    if (!stuck && wdone == (int)vslots) {
                      const int start = row_get(last_base + partition_idx * r_chs + c);
                      if (PM(1)) recs[(uint32_t)start + ((V.y >> 24) & 7u)] = make_uint2((V.z & 0xFFFF0000u) | nent, V.w | ((uint32_t)c << 25));
                      nent += vslots;
                      ++nops;
                      continue;
                    }
                    fast_done = wdone;
                  }
                }
                if ((r.cascade[cls] & (1 << stage)) == 0) continue;
                const int book_idx = r.books[cls][stage];
                if (book_idx < 0) continue;
                const NvhPBook book = books[book_idx];  // by value (see class_book)
                const int dims = book.dims;
                if (dims == 0) {
                  err = kErrRuntime;
                  break;
                }
                if (partition_idx > 0xFFFF) {
                  err = kErrUnsupported;
                  break;
                }
                NvhResOp op;
                op.ent_off = ent_base + nent;
                op.partition = (uint16_t)partition_idx;
                op.channel = (uint8_t)c;
                op.book = (uint8_t)book_idx;
                bool push = false, bad = false;
                if (r_type == 0) {
                  // Residue0.WriteVectors (:180-201): decode all entries first, add only if all decoded
                  const int steps = r_psize / dims;
                  const uint32_t mark = nent;
                  if (nent + (uint32_t)steps > (uint32_t)T.cap_ent) {
                    err = kErrRuntime;
                    break;
                  }
                  for (int i = 0; i < steps; i++) {
                    const int e = decode_scalar<LDS, UNI>(T, s_prefix, s_pkt, book, p);
                    if (e == -2) {
                      err = kErrRuntime;
                      break;
                    }
                    if (e == -1) {
                      bad = true;
                      break;
                    }
                    entries[ent_base + nent++] = (uint16_t)e;
                  }
                  if (err) break;
                  if (bad) {
                    nent = mark;
                    stop = true;
                    stop_p = partition_idx; stop_c = c; stop_pushed = false;
                    break;
                  }
                  if (offset + steps * dims > buflen) {
                    err = kErrRuntime;
                    break;
                  }
                  push = true;
                } else {
                  // Residue1.WriteVectors (Residue1.cs:8-26) / Residue2.WriteVectors (Residue2.cs:23-47):
                  // vectors are added as they are decoded; a failed decode keeps what was added so far
                  // (partition_size + dims - 1) / dims by the book's reciprocal (exact: nvh_setup.hip checked the range)
                  const int slots = dims > 1 ? (int)__umulhi((uint32_t)(r_psize + dims - 1), book.dim_magic) : r_psize;
                  if (nent + (uint32_t)slots > (uint32_t)T.cap_ent) {
                    err = kErrRuntime;
                    break;
                  }
                  int done = fast_done;  // (UNI: entries the visit's fast form above has decoded already)
                  PT_ACC_BEGIN();
                  // The vector's entries, fast form: while the packet has 32 bits left, the code resolves in the book's LDS prefix
                  // table and slots remain, a symbol is a masked ds_read, a 64-bit shift and a 16-bit store in a loop with one
                  // exit test.  (The general loop below, with the reference's end-of-packet and null-list rules inlined into a
                  // four-deep nest of divergent loops, compiled to some hundred mostly scalar exec-mask instructions per symbol:
                  // 145 M SALU + 127 M VALU per 4096 packets.)  Whatever this loop leaves -- the last bits of a packet, a long
                  // code, a book whose table lies in global memory -- the general loop takes up where it stopped.
                  if (book.has_tree && book.lds_off != 0xFFFFFFFFu) {
                    const uint32_t pmask = (1u << book.prefix_bits) - 1u, toff = book.lds_off;
                    uint16_t* __restrict__ eout = entries + ent_base + nent;
                    // (one exit test: the table read is always in range, so it is not guarded)
                    uint32_t node = (UNI && LDS) ? 0u : s_prefix[toff + ((uint32_t)p.buf & pmask)];  // (UNI: read where it is needed)
                    auto consume = [&](const uint32_t len) {
                      p.buf >>= len;
                      p.avail -= len;
                      p.pos += len;
                      if (p.avail <= 32u && p.next < p.nwords) {  // (one word restores br_fill's invariant: len <= 32)
                        const uint32_t word = p.ahead;
                        p.next++;
                        p.ahead = br_word<LDS>(p, s_pkt, p.next);
                        p.buf |= (uint64_t)word << p.avail;
                        p.avail += 32u;
                      }
                    };
                    for (;;) {
                      if constexpr (UNI && LDS) {
                        // (decode_windows above: the vector's entries by windows of up to 63 start positions)
                        const bool stuck = decode_windows(toff, pmask, eout, slots, done);
                        if (stuck || !((int)(done < slots) & (int)(p.total - p.pos >= 32u) & (int)(book.has_overflow != 0))) break;
                        node = s_prefix[toff + ((uint32_t)p.buf & pmask)];
                        if (node & 0x80u) continue;  // (cannot happen: a window ends in front of a code only if it does not resolve)
                      } else {
                      while ((bool)((int)(done < slots) & (int)(p.total - p.pos >= 32u) & (int)((node >> 7) & 1u))) {
                        consume(node & 0x7Fu);
                        eout[done] = (uint16_t)(node >> 8);
                        ++done;
                        node = s_prefix[toff + ((uint32_t)p.buf & pmask)];
                      }
                      }
                      // A code longer than the prefix (Codebook.cs:307-318), still 32 bits left: its slot's group of overflow nodes
                      // (decode_scalar's scan, same nodes in the same order) without leaving this loop -- the general loop below
                      // costs some hundred exec-mask instructions per symbol, and packets that use the long codes often (the C5
                      // writer's, which draws entries uniformly) spent most of their parse there.
                      if (!((int)(done < slots) & (int)(p.total - p.pos >= 32u) & (int)(book.has_overflow != 0))) break;
                      const uint32_t data = (uint32_t)p.buf & (book.max_bits >= 32 ? 0xFFFFFFFFu : (1u << book.max_bits) - 1u);
                      uint32_t cnt = node & 0x7Fu;
                      uint32_t hit_len = 0, hit_val = 0;
                      if (cnt != 0x7Fu && book.ovf_lds != 0xFFFFFFFFu) {
                        // the group's nodes from LDS (nvh_setup.hip: 8 bytes each): a probe is an LDS round trip, not an L2 one
                        const uint32_t g = book.ovf_lds + 2u * (node >> 8);
                        for (uint32_t k = 0; k < cnt; ++k) {
                          const uint32_t bits = s_prefix[g + 2u * k], vl = s_prefix[g + 2u * k + 1u], len = vl & 0xFFu;
                          if (bits == (data & ((1u << len) - 1u))) {
                            hit_val = vl >> 8;
                            hit_len = len;
                            break;
                          }
                        }
                      } else {
                        const NvhPOverflow* __restrict__ ov = T.overflow + book.ovf_off;
                        if (cnt == 0x7Fu) cnt = book.ovf_count;
                        else ov += book.ovf_count + (node >> 8);
                        for (uint32_t k = 0; k < cnt; ++k) {
                          const uint4 o = *reinterpret_cast<const uint4*>(ov + k);  // bits, mask, value, length
                          if (o.x == (data & o.y)) {
                            hit_val = o.z;
                            hit_len = o.w;
                            break;
                          }
                        }
                      }
                      if (hit_len == 0u || hit_len > 32u) break;  // no match (or nothing this loop may skip): the general loop decides
                      consume(hit_len);
                      eout[done] = (uint16_t)hit_val;
                      ++done;
                      node = s_prefix[toff + ((uint32_t)p.buf & pmask)];
                    }
                    nent += (uint32_t)done;
                  }
                  for (int i = done * dims; i < r_psize; i += dims) {
                    const int e = decode_scalar<LDS, UNI>(T, s_prefix, s_pkt, book, p);
                    if (e == -2) {
                      err = kErrRuntime;
                      break;
                    }
                    if (e == -1) {
                      bad = true;
                      break;
                    }
                    entries[ent_base + nent++] = (uint16_t)e;
                    ++done;
                  }
                  PT_ACC_END(0);
                  if (err) break;
                  if (done > 0) {  // bounds of the adds the reference performed
                    const int last = done * dims - 1;
                    if (r_type == 1) {
                      if (offset + last >= buflen) err = kErrRuntime;
                    } else {
                      // offset / real_channels + last / real_channels (Residue2.cs:27, :30-45) by the residue's reciprocal
                      const int ob = r_rch > 1 ? (int)__umulhi((uint32_t)offset, r_rchm) : offset;
                      const int lb = r_rch > 1 ? (int)__umulhi((uint32_t)last, r_rchm) : last;
                      if (ob + lb >= buflen) err = kErrRuntime;
                    }
                    if (err) break;
                  }
                  for (int i = done; i < slots; i++) entries[ent_base + nent++] = (uint16_t)NVH_ENTRY_SKIP;
                  push = true;
                }
                if (push) {
                  if (nops >= (uint32_t)T.cap_ops) {
                    err = kErrRuntime;
                    break;
                  }
                  if constexpr (SLAB) {
                    const unsigned cm = r.book_mask[cls];
                    const int start = row_get(last_base + partition_idx * r_chs + c);
                    const unsigned rank = (unsigned)__popc(cm & ((1u << stage) - 1u));
                    const uint32_t rw[2] = {NVH_SLAB_REC(op.ent_off - ent_base, book.slab_dm16, book.slab_lat & 0xFFFFu, book.slab_lat >> 16, dims, c,
                                                         stage, (cm >> (stage + 1)) != 0)};
                    if (PM(1)) recs[(uint32_t)start + rank] = make_uint2(rw[0], rw[1]);
                  } else {
                    const uint32_t rel = nops;
                    ops[op_base + rel] = op;
                    uint16_t lk = (uint16_t)NVH_LINK_NONE;
                    if (rel >= (uint32_t)NVH_LINK_NONE) links_ok = false;
                    const int last_i = last_base + partition_idx * r_chs + c;
                    const int last = row_get(last_i);
                    if (last >= 0 && rel < (uint32_t)NVH_LINK_NONE) {
                      op_link[op_base + (uint32_t)last] = (uint16_t)((op_link[op_base + (uint32_t)last] & 0x8000u) | (uint16_t)rel);
                      lk |= 0x8000u;
                    }
                    op_link[op_base + rel] = lk;
                    row_set(last_i, (int)rel);
                  }
                  ++nops;
                }
                if (bad) {
                  stop = true;
                  stop_p = partition_idx; stop_c = c; stop_pushed = true;
                  break;
                }
              }
            }
          }
        }
        }  // (the nest)
        if (SLAB && stop && !err) {
          // The packet ran out (Residue0.cs:160-168): the vector writes behind that point never happened, but their records are
          // part of chains that were allocated by class -- they get entries that say "no vector" (quirks B-14 / B-16), so that the
          // synthesis kernel needs no notion of a truncated chain.  `stage` is one past the stage the packet ended in.
          const int stop_stage = stage - 1;
          for (int pi = 0; pi < partition_count && !err; ++pi)
            for (int c = 0; c < r_chs && !err; ++c) {
              const int start = row_get(last_base + pi * r_chs + c);
              if (start < 0) continue;
              const int cls = row_get(c * pw_stride + pi);
              const unsigned cm = r.book_mask[cls];
              for (int st = 0; st < r_stages; ++st) {
                if (!((cm >> st) & 1u)) continue;
                const bool written = st < stop_stage ||
                                     (st == stop_stage && (pi < stop_p || (pi == stop_p && (c < stop_c || (c == stop_c && stop_pushed)))));
                if (written) continue;
                const NvhPBook book = books[r.books[cls][st]];
                const int dims = book.dims;
                const int slots = r_type == 0 ? r_psize / dims : (r_psize + dims - 1) / dims;
                if (nent + (uint32_t)slots > (uint32_t)T.cap_ent) {
                  err = kErrRuntime;
                  break;
                }
                const uint32_t rw[2] = {NVH_SLAB_REC(nent, book.slab_dm16, book.slab_lat & 0xFFFFu, book.slab_lat >> 16, dims, c, st, (cm >> (st + 1)) != 0)};
                recs[(uint32_t)start + (unsigned)__popc(cm & ((1u << st) - 1u))] = make_uint2(rw[0], rw[1]);
                for (int i = 0; i < slots; ++i) entries[ent_base + nent++] = (uint16_t)NVH_ENTRY_SKIP;
              }
            }
        }
        s_rtype = r_type; s_rch = r_rch; s_psz = r_psize; s_rbegin = r_begin;
        s_parts = err ? 0 : partition_count; s_chs = r_chs; s_b1 = r.alias_b1;
      }
      if (SLAB) {
        if (!ran) { s_rtype = r.type; s_rch = r.real_channels; s_psz = r.partition_size; s_rbegin = r.begin; s_b1 = r.alias_b1; s_parts = 0; }
        s_res = residue_idx;
        if (gen_rows) {
          if (s_npass >= T.cap_pass) {
            err = kErrRuntime;
          } else {
            int* rows = LDS ? l_rows_base : g_rows_base;
            rows[T.cap_parts + s_npass * row_stride] = residue_idx;
            rows[T.cap_parts + s_npass * row_stride + 1] = (ran && !err) ? s_parts : 0;
          }
        }
        s_npass += 1;
      }
      // stages not reached keep empty ranges
      if (!err) {
        for (int s = ran ? stage : 0; s <= NVH_MAX_STAGES; s++) pass.op_begin[s] = op_base + nops;
        if (npass >= (uint32_t)T.cap_pass) {
          err = kErrRuntime;
        } else {
          passes[pass_base + npass++] = pass;
        }
      }
    }
    if (!err) {
      exec_mask = (force_e | energy) & ~force_no;
      for (int c = 0; c < nch; c++) ch_out[c].exec = (exec_mask >> c) & 1u;
    }
  }

  PT_T(3);
#ifdef NVH_DEBUG
  if (PHASE != 2 && dbg && active) { dbg[(long long)f * 24 + 8] = pt_acc[0]; dbg[(long long)f * 24 + 9] = pt_acc[1]; dbg[(long long)f * 24 + 10] = pt_acc[2]; dbg[(long long)f * 24 + 11] = pt_rounds; }
#endif
  if constexpr (PHASE == 1) {
    // (every store of the parse -- posts, rows, records, entries, channel records -- is in global memory: the tail kernel's)
    uint4* const hv = reinterpret_cast<uint4*>(ho);
    hv[0] = make_uint4((uint32_t)err, nrec_alloc, nent, nops);
    hv[1] = make_uint4(npass, exec_mask, (uint32_t)pcs, (uint32_t)(pcs >> 32));
    hv[2] = make_uint4((uint32_t)s_rtype, (uint32_t)s_rch, (uint32_t)s_psz, (uint32_t)s_rbegin);
    hv[3] = make_uint4((uint32_t)s_npass, (uint32_t)s_parts, (uint32_t)s_chs | ((uint32_t)s_b1 << 8) | ((links_ok ? 1u : 0u) << 16), (uint32_t)s_res);
    return;
  }
  if constexpr (UNI) __threadfence_block();  // the entries the window decode's lanes stored are read by the packet's lane below
  uint32_t slab_vecs = 0;
  if constexpr (SLAB) {
    // ---- behind the parse, the lanes of the wavefront side by side: the rest of the slab ----
    NvhSlabHdr H;
    H.n = 0; H.exec_mask = 0; H.flags = 0; H.nheads = 0; H.nrec = 0;
    H.off_heads = H.off_rec = H.off_ent = NVH_SLAB_HDR_VECS; H.vecs = NVH_SLAB_HDR_VECS;
    H.lpc = 0; H.rgeom = 0; H.group = 2; H.lpc_magic = 0; H.frame = (uint32_t)f; H.coupling = 0;
    for (int c = 0; c < NVH_SLAB_MAX_CH; ++c) H.chan[c] = (uint32_t)NVH_SLAB_HDR_VECS << 16;
    uint32_t off = NVH_SLAB_HDR_VECS, b1_bins = 0, gen_list = 0;
    bool fault = false, gen_frame = false;
    const bool mine = active && fr.n != 0 && !err;  // this lane has a frame to finish
    if (mine) {
      H.n = (uint16_t)fr.n;
      H.exec_mask = (uint8_t)(exec_mask & 0xFFu);
      if (fr.mdct_slot) H.flags |= NVH_SLAB_MDCT_SLOT;
      // records (written during the parse) | heads | entries | floors
      const uint32_t off_rec = NVH_SLAB_HDR_VECS;
      const uint32_t off_heads = off_rec + ((nrec_alloc + 1) >> 1);
      if (nrec_alloc & 1u) recs[nrec_alloc] = make_uint2(0u, 0u);
      auto row_get = [&](int i) { return LDS ? l_rows_base[i] : g_rows_base[i]; };
      const int row0 = gen_rows ? T.cap_parts + 2 : T.cap_parts;  // chain-start row of the (first) pass
      // a frame of the general bin walk (host_slab.cpp: build_slabs): more than one pass, or a residue outside the pair path and B-1
      gen_frame = gen_rows && (s_npass > 1 || (s_npass == 1 && residues[s_res].general != 0));
      if (!gen_frame) {
        // the chain heads, from the rows of the walk: first record | the partition's first bin << 16, partition by partition
        uint32_t* hd = reinterpret_cast<uint32_t*>(slab + off_heads);
        const unsigned rchm = s_rch > 1 ? (unsigned)((0x100000000ull + (unsigned)s_rch - 1) / (unsigned)s_rch) : 0u;
        for (int pi = 0; pi < (PM(2) ? s_parts : 0); ++pi) {
          const unsigned offset = (unsigned)s_rbegin + (unsigned)pi * (unsigned)s_psz;
          const unsigned xb0 = (s_rtype == 2 && s_rch > 1) ? __umulhi(offset, rchm) : offset;
          for (int c = 0; c < s_chs; ++c) {
            const int start = row_get(row0 + pi * s_chs + c);
            if (start >= 0) hd[nheads++] = (uint32_t)start | (xb0 << 16);
          }
          if (xb0 > 0xFFFFu) err = kErrUnsupported;
        }
        for (uint32_t i = nheads; i < ((nheads + 3) & ~3u); ++i) hd[i] = 0;
      } else {
        // every pass's chains, pass by pass (the group list below numbers them in this very order)
        uint32_t* hd = reinterpret_cast<uint32_t*>(slab + off_heads);
        for (int k = 0; k < s_npass; ++k) {
          const int blk = T.cap_parts + k * row_stride;
          const NvhPResidue& rk = residues[row_get(blk)];
          const int np = row_get(blk + 1), chs = rk.channels;
          const unsigned rch = rk.type == 2 ? (unsigned)rk.real_channels : 1u;
          const unsigned rchm = rch > 1 ? (unsigned)((0x100000000ull + rch - 1) / rch) : 0u;
          for (int pi = 0; pi < np; ++pi) {
            const unsigned offset = (unsigned)rk.begin + (unsigned)pi * (unsigned)rk.partition_size;
            const unsigned xb0 = rch > 1 ? __umulhi(offset, rchm) : offset;
            for (int c = 0; c < chs; ++c) {
              const int start = row_get(blk + 2 + pi * chs + c);
              if (start >= 0) hd[nheads++] = (uint32_t)start | (xb0 << 16);
            }
            if (xb0 > 0xFFFFu) err = kErrUnsupported;
          }
        }
        for (uint32_t i = nheads; i < ((nheads + 3) & ~3u); ++i) hd[i] = 0;
        if (nheads > 0xFFFEu) err = kErrUnsupported;
      }
      const uint32_t off_ent = off_heads + ((nheads + 3) >> 2);
      off = off_ent + ((nent + 7) >> 3);
      if (gen_frame) {
        // the group list (nvh_format.h: NvhSlabHdr::group == 1; host_slab.cpp: residue_general): count | per (pass, channel) of a
        // per-channel residue, per pass of a Residue2: two units of geometry | uint16 pchain[]: the chain of every partition
        uint32_t* w = reinterpret_cast<uint32_t*>(slab + off);
        uint32_t ngroups = 0, npc = 0;
        for (int k = 0; k < s_npass; ++k) ngroups += (uint32_t)residues[row_get(T.cap_parts + k * row_stride)].channels;
        w[0] = ngroups; w[1] = w[2] = w[3] = 0;
        uint16_t* pc = reinterpret_cast<uint16_t*>(w + 4 + 8 * ngroups);
        uint32_t gi = 0, hk = 0;
        for (int k = 0; k < s_npass; ++k) {
          const int blk = T.cap_parts + k * row_stride;
          const NvhPResidue& rk = residues[row_get(blk)];
          const int np = row_get(blk + 1), chs = rk.channels;
          const unsigned rch = rk.type == 2 ? (unsigned)rk.real_channels : 1u, psz = (unsigned)rk.partition_size;
          for (int c = 0; c < chs; ++c) {
            uint32_t* q = w + 4 + 8 * gi;
            q[0] = (uint32_t)rk.begin; q[1] = psz; q[2] = (uint32_t)np; q[3] = (rk.span_max + rch - 1u) / rch;
            q[4] = (uint32_t)rk.type | (rch << 4) | ((uint32_t)k << 8) | ((uint32_t)(rk.type == 2 ? 0 : c) << 12);
            q[5] = (uint32_t)((0x100000000ull + psz - 1) / psz);
            q[6] = npc + (uint32_t)(c * np);
            q[7] = 0;
            ++gi;
          }
          // chain numbers in the order the heads were written: partition by partition, channel by channel
          for (int pi = 0; pi < np; ++pi)
            for (int c = 0; c < chs; ++c) pc[npc + (uint32_t)(c * np + pi)] = row_get(blk + 2 + pi * chs + c) >= 0 ? (uint16_t)hk++ : (uint16_t)0xFFFFu;
          npc += (uint32_t)(chs * np);
        }
        for (uint32_t i = npc; i < ((npc + 7u) & ~7u); ++i) pc[i] = 0xFFFFu;
        if (npc > 0xFFFFFFu) err = kErrUnsupported;
        gen_list = off;
        off += 1u + 2u * ngroups + ((npc + 7u) >> 3);
      } else if (s_b1) {
        // quirk B-1 (kernels_synth.hip: residue_walk_bins): the residue's geometry and the chain of every partition -- the heads
        // above are in partition order, one per partition that has a chain (Residue2: a single channel)
        uint32_t* prm = reinterpret_cast<uint32_t*>(slab + off);
        prm[0] = (uint32_t)s_rbegin; prm[1] = (uint32_t)s_psz; prm[2] = (uint32_t)s_parts;
        prm[3] = ((uint32_t)s_psz + (uint32_t)s_rch - 1u) / (uint32_t)s_rch;
        uint16_t* pchain = reinterpret_cast<uint16_t*>(prm + 4);
        uint32_t k = 0;
        for (int pi = 0; pi < s_parts; ++pi) pchain[pi] = row_get(row0 + pi) >= 0 ? (uint16_t)k++ : (uint16_t)0xFFFFu;
        for (int pi = s_parts; pi < ((s_parts + 7) & ~7); ++pi) pchain[pi] = 0xFFFFu;
        b1_bins = off;
        off += 1u + (uint32_t)((s_parts + 7) >> 3);
      }
      {
        // the frame's entries from their per-frame area (16-byte aligned: cap_ent is a multiple of eight), the tail padded with
        // "no vector"
        const uint4* src = reinterpret_cast<const uint4*>(entries + ent_base);
        uint4* dst = slab + off_ent;
        const uint32_t full = PM(4) ? nent >> 3 : 0;
        if constexpr (PHASE != 2) {
          for (uint32_t i = 0; i < full; ++i) dst[i] = src[i];
        }  // (PHASE 2: by the whole wavefront, below)
        if (nent & 7u) {
          uint16_t* d16 = reinterpret_cast<uint16_t*>(dst + full);
          const uint16_t* s16 = entries + ent_base + 8 * full;
          for (uint32_t i = 0; i < 8; ++i) d16[i] = i < (nent & 7u) ? s16[i] : (uint16_t)NVH_ENTRY_SKIP;
        }
      }
      H.nheads = (uint16_t)nheads;
      H.nrec = (uint16_t)nrec_alloc;
      H.off_rec = (uint16_t)off_rec;
      H.off_heads = (uint16_t)off_heads;
      H.off_ent = (uint16_t)off_ent;
    }
    if constexpr (PHASE == 2) {
      // the frame's entries, 16 bytes per lane (the packet's lane has done the last partial vector)
      const uint32_t oe = (uint32_t)__shfl(mine ? (int)H.off_ent : 0, 0);
      if (__shfl((int)mine, 0)) {
        const uint4* src = reinterpret_cast<const uint4*>(entries + ent_base);
        uint4* dst = slab + oe;
        const uint32_t full = nent >> 3;
        for (uint32_t i = (uint32_t)lane; i < full; i += 64u) dst[i] = src[i];
      }
    }
    PT_T(4);
    // ---- floors, the wavefront together: the packets of its lanes one after the other, lane = post (floor_to_slab_wave) ----
    {
      // one floor scratch block per wavefront behind the per-lane areas (16-byte aligned), then one error word each
      const int nwaves = (int)(blockDim.x >> 6);
      const int fs_word = (tab_words + T.meta_words + nwaves * lanes * (scratch_words + pkt_words) + 3) & ~3;
      FloorScratch* Q = reinterpret_cast<FloorScratch*>(s_prefix + fs_word) + wave;
      int* s_err = reinterpret_cast<int*>(s_prefix + fs_word + nwaves * NVH_SP_FLOOR_SCRATCH_WORDS) + wave;
      for (int j = 0; j < (PM(8) ? lanes : 0); ++j) {
        const int jn = __shfl(mine ? fr.n : 0, j);
        if (jn == 0) continue;  // uniform
        const unsigned jexec = (unsigned)__shfl((int)exec_mask, j);
        const unsigned jpl = (unsigned)__shfl((int)(unsigned)(pcs & 0xFFFFFFFFull), j), jph = (unsigned)__shfl((int)(unsigned)(pcs >> 32), j);
        const unsigned long long jpcs = ((unsigned long long)jph << 32) | jpl;
        const int jf = __shfl(f, j), jmap = __shfl(fr.mapping, j);
        unsigned joff = (unsigned)__shfl((int)off, j);
        const NvhPMapping& map = mappings[jmap];
        uint4* jslab = slabs + (long long)jf * T.slab_stride_vecs;
        const int half = jn >> 1;
        for (int c = 0; c < nch; ++c) {
          const int pc = (int)((jpcs >> (7 * c)) & 0x7Fu);
          const int mode = ((jexec >> c) & 1u) ? (pc > 0 ? 1 : 2) : 0;
          int ns = 0;
          const unsigned coff = joff;
          if (mode == 1) {
            if (lane == 0) *s_err = 0;
            sp_wave_sync();
            // (room for the segments: at most posts + 1 of them, plus the table -- the stride is the setup's worst case)
            ns = floor_to_slab_wave(Q, &T.dfloors[map.chan_floor[c]].f1, posts + ((long long)jf * nch + c) * NVH_MAX_POSTS, pc, half, T.recip, s_err,
                                    jslab + joff, lane);
            joff += (unsigned)ns + (unsigned)(((half >> 2) + 15) >> 4);
          }
          if (lane == j) {
            H.chan[c] = (uint32_t)mode | ((uint32_t)ns << 8) | (coff << 16);
            if (mode == 1 && *s_err) fault = true;
          }
        }
        if (lane == j) off = joff;
      }
    }
    if (mine) {
      const NvhPMapping& map = mappings[fr.mapping];
      if (fault) H.flags |= NVH_SLAB_FLOOR_FAULT;
      // residue geometry (Residue0.cs:157-170, Residue2.cs:23-47): components a lane of the synthesis kernel owns
      if (gen_frame) {
        H.group = 1;
        H.lpc = (uint16_t)gen_list;
        H.lpc_magic = 0;
      } else if (s_npass == 1) {
        unsigned group = ((unsigned)s_psz & 7u) == 0 ? 8u : 2u;
        if (s_rtype == 2 && s_rch > 2) group = ((unsigned)s_psz % (2u * (unsigned)s_rch)) == 0 ? 2u * (unsigned)s_rch : 2u;  // (host: slab setups only)
        const unsigned lpc = (unsigned)s_psz / group;
        H.group = (uint8_t)group;
        H.lpc = (uint16_t)lpc;
        H.lpc_magic = lpc > 1 ? (uint32_t)((0x100000000ull + lpc - 1) / lpc) : 0u;
        if (s_b1) {  // the bin walk: group 0, the section's offset and the partition size's reciprocal in the lane-count fields
          H.group = 0;
          H.lpc = (uint16_t)b1_bins;
          H.lpc_magic = (uint32_t)((0x100000000ull + (unsigned)s_psz - 1) / (unsigned)s_psz);
        }
      }
      H.rgeom = gen_frame ? (uint8_t)(1 | (1 << 4)) : (uint8_t)(s_rtype | (s_rch << 4));
      // inverse coupling (Mapping.cs:137-182): in the chain walk when one lane holds both channels of a bin, else passes
      if (nch == 2 && map.coupling_steps == 1 && s_npass == 1 && s_rtype == 2 && s_rch == 2 && !gen_frame) {
        if ((exec_mask & 3u) != 0) {
          if (map.coupling_mag[0] == 1) H.flags |= NVH_SLAB_MG1;
          H.flags |= NVH_SLAB_SWEEP_COUPLES;
        }
      } else if (map.coupling_steps > 0) {
        unsigned word = 0, cnt = 0;
        for (int st = map.coupling_steps - 1; st >= 0; --st) {
          const unsigned mg = map.coupling_mag[st], an = map.coupling_ang[st];
          if (((exec_mask >> mg) | (exec_mask >> an)) & 1u) {
            word |= (mg | (an << 3)) << (4 + 6 * cnt);
            ++cnt;
          }
        }
        if (cnt) {
          H.coupling = word | cnt;
          H.flags |= NVH_SLAB_COUPLE_PASS;
        }
      }
      if (nch <= 2 && !(H.flags & NVH_SLAB_COUPLE_PASS) &&
          (s_npass == 0 || (!gen_frame && H.group == 8 && ((unsigned)s_rbegin & ((s_rtype == 2 && s_rch == 2) ? 7u : 3u)) == 0)))
        H.flags |= NVH_SLAB_FUSE_FLOOR;
      if (off > (uint32_t)T.slab_stride_vecs || off > 0xFFFFu) err = kErrRuntime;
      H.vecs = (uint16_t)off;
    }
    if (active && !err) {
      // (paired emission is entered into the header by k_parse_links, which knows the neighbours' execute flags)
      const uint4* hv = reinterpret_cast<const uint4*>(&H);
      for (int i = 0; i < NVH_SLAB_HDR_VECS; ++i) slab[i] = hv[i];
      slab_vecs = H.vecs;
    }
  }
  PT_T(5);
  if (!active) return;
  frames[f].pass_begin = pass_base;
  frames[f].pass_end = pass_base + npass;
  frames[f].op_begin = op_base;
  frames[f].op_count = nops;
  frames[f].ent_begin = ent_base;
  frames[f].ent_count = nent;
  frames[f].exec_mask = exec_mask;
  // batch maxima: look before the atomic -- they settle after a few packets, and tens of thousands of atomics on one
  // cache line are not free
  if ((int)nops > __atomic_load_n(&result->max_ops, __ATOMIC_RELAXED)) atomicMax(&result->max_ops, (int)nops);
  if ((int)nent > __atomic_load_n(&result->max_ent, __ATOMIC_RELAXED)) atomicMax(&result->max_ent, (int)nent);
  if ((int)npass > __atomic_load_n(&result->max_pass, __ATOMIC_RELAXED)) atomicMax(&result->max_pass, (int)npass);
  if (SLAB && (int)slab_vecs > __atomic_load_n(&result->max_vecs, __ATOMIC_RELAXED)) atomicMax(&result->max_vecs, (int)slab_vecs);
  if (!links_ok) atomicAnd(&result->links_ok, 0);
  if (err) {
    const int prev = atomicMin(&result->err_frame, f);
    if (f < prev) result->err_code = err;  // benign race between several failing frames: the host re-checks the minimum
  }
}

#define NVH_PARSE_KERNEL(NAME, LDSV, SLABV, UNIV, CURV, PHASEV)                                                                          \
  extern "C" __global__ void __launch_bounds__(64 * NVH_PARSE_MAX_WAVES)                                                                  \
  NAME(NvhDevParse T, const uint8_t* __restrict__ pkt_pool, const NvhPacketRef* __restrict__ refs, int nframes,                          \
       NvhFrame* __restrict__ frames, NvhChan* __restrict__ chans, NvhResPass* __restrict__ passes, NvhResOp* __restrict__ ops,          \
       uint16_t* __restrict__ op_link, uint16_t* __restrict__ entries, uint16_t* __restrict__ posts, int* __restrict__ scratch,          \
       NvhParseResult* __restrict__ result, int lanes, int scratch_words, int pkt_words, uint4* __restrict__ slabs,                     \
       const int* __restrict__ order, uint32_t* __restrict__ handover NVH_DBG_PARAMS) {                                                 \
    parse_body<LDSV, SLABV, UNIV, CURV, PHASEV>(T, pkt_pool, refs, nframes, frames, chans, passes, ops, op_link, entries, posts, scratch, \
                                                result, lanes, scratch_words, pkt_words, slabs, order, handover NVH_DBG_ARGS);           \
  }
NVH_PARSE_KERNEL(k_parse, true, false, false, false, 0)       // descriptors out, packets and scratch rows in LDS
NVH_PARSE_KERNEL(k_parse_g, false, false, false, false, 0)    // ... in global memory (a packet too long for the LDS budget)
NVH_PARSE_KERNEL(k_parse_slab, true, true, false, false, 0)   // slabs out (the stream shapes the slab synthesis kernels take)
NVH_PARSE_KERNEL(k_parse_slab_g, false, true, false, false, 0)
NVH_PARSE_KERNEL(k_parse_slab_u, true, true, true, false, 0)  // k_parse_slab for one packet per wavefront, wave-uniform (UNI above)
// several packets per wavefront, every lane with its own cursor through the residue walk (CUR above), in two kernels (PHASE above):
// the parse with up to 64 packets per wavefront, then the rest of the slab with one wavefront per packet
NVH_PARSE_KERNEL(k_parse_slab_c, false, true, false, true, 1)
NVH_PARSE_KERNEL(k_parse_slab_t, false, true, false, false, 2)

// ---- the lean form of the multi-packet parse (k_parse_slab_f) ----
// What k_parse_slab_c's measurements said (profiles/r06_cursor.txt): with the lanes' vectors decoded side by side the walk was
// no faster on real packets, and neither its rows in global memory nor its stores per symbol were why -- moving both into LDS
// changed nothing.  The general body compiles to ~13 k instructions with its condition masks spilled from SGPRs into VGPR lanes:
// a step of the cursor costs ~1.7 k cycles of exec-mask bookkeeping.  This kernel is the same walk with nothing in it that the
// ordinary packet does not need: one residue pass, Residue1 / Residue2, every vector of a visit in windows of the book's LDS
// prefix table (long codes from their slot's group in LDS), whole packets.  Whatever else turns up -- a packet that ends inside
// the residue, a book outside the LDS image, a Residue0, several submaps, a fault the reference would throw on -- is not decided
// here: the lane marks its frame (NVH_PHO_BAIL) and k_parse_slab_c, launched behind this kernel over the marked frames only,
// parses that packet from its first bit.  Same reads in the same order, same records, entries and rows as the general body for
// every packet this kernel completes; the tail kernel (k_parse_slab_t) does not know which of the two wrote them.
extern "C" __global__ void __launch_bounds__(512)
k_parse_slab_f(NvhDevParse T, const uint8_t* __restrict__ pkt_pool, const NvhPacketRef* __restrict__ refs, int nframes,
               NvhFrame* __restrict__ frames, NvhChan* __restrict__ chans, NvhResPass* __restrict__ passes, NvhResOp* __restrict__ ops,
               uint16_t* __restrict__ op_link, uint16_t* __restrict__ entries, uint16_t* __restrict__ posts, int* __restrict__ scratch,
               NvhParseResult* __restrict__ result, int lanes, int scratch_words, int pkt_words, uint4* __restrict__ slabs,
               const int* __restrict__ order, uint32_t* __restrict__ handover NVH_DBG_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_prefix[];
  uint32_t* s_meta = s_prefix + T.lds_words;
  for (int i = threadIdx.x; i < T.lds_words; i += (int)blockDim.x) s_prefix[i] = T.lds_image[i];
  {
    const uint32_t* gm = reinterpret_cast<const uint32_t*>(T.books);
    for (int i = threadIdx.x; i < T.meta_words; i += (int)blockDim.x) s_meta[i] = gm[i];
  }
  // (scratch_words: the words of the second-level image the host found room for -- all of it or none)
  uint32_t* const s_sub = s_meta + T.meta_words;
  const int sub_words = scratch_words;
  for (int i = threadIdx.x; i < sub_words; i += (int)blockDim.x) s_sub[i] = T.sub_image[i];
  __syncthreads();
  const NvhPBook* books = reinterpret_cast<const NvhPBook*>(s_meta);
  const NvhPFloor1* floors = reinterpret_cast<const NvhPFloor1*>(reinterpret_cast<const uint8_t*>(s_meta) + T.meta_floors_off);
  const NvhPResidue* residues = reinterpret_cast<const NvhPResidue*>(reinterpret_cast<const uint8_t*>(s_meta) + T.meta_residues_off);
  const NvhPMapping* mappings = reinterpret_cast<const NvhPMapping*>(reinterpret_cast<const uint8_t*>(s_meta) + T.meta_mappings_off);
  uint16_t* const s_stage16 = reinterpret_cast<uint16_t*>(s_sub + sub_words);                          // NVH_PSTG entries per lane
  // (the per-lane rows are indexed by packet of the workgroup -- nt of them, this lane's is tid --, not by thread: see parse_body)
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int nt = (int)(blockDim.x >> 6) * lanes, tid = wave * lanes + lane;
  uint8_t* const s_cls = reinterpret_cast<uint8_t*>(s_stage16) + (size_t)nt * (NVH_PSTG * 2);  // a byte per (channel, partition)
  const int f_idx = (int)blockIdx.x * nt + tid;
  if (lane >= lanes || f_idx >= nframes) return;
  const int f = order ? order[f_idx] : f_idx;
  const NvhFrame fr = frames[f];
  const int nch = T.channels;
  NvhChan* ch_out = chans + (long long)f * nch;
  const uint32_t ent_base = (uint32_t)f * (uint32_t)T.cap_ent;
  uint2* const recs = reinterpret_cast<uint2*>(slabs + (long long)f * T.slab_stride_vecs + NVH_SLAB_HDR_VECS);
  int* const g_rows = scratch + (long long)f * T.row_words;
  bool bail = false;
#ifdef NVH_DEBUG
  long long ft[4] = {clock64(), 0, 0, 0}, facc[3] = {0, 0, 0}, frounds = 0;  // stamps: start, floors done, residue done; cycles in: steps, entries, flush + record
  int why = 0;  // profiling builds: which test left the packet to the general body (tools/dbg_phase_parse.py prints the histogram)
#define NVH_WHY(k) why = (k)
#else
#define NVH_WHY(k) do { } while (0)
#endif
  uint32_t nrec_alloc = 0, nent = 0, nops = 0, npass = 0, exec_mask = 0;
  unsigned long long pcs = 0;
  int s_rtype = 0, s_rch = 1, s_psz = 0, s_rbegin = 0, s_npass = 0, s_parts = 0, s_chs = 1, s_b1 = 0, s_res = 0;
  if (fr.n != 0) {
    const NvhPacketRef ref = refs[f];
    BitR p;
    br_init<false>(p, reinterpret_cast<const uint32_t*>(pkt_pool + ref.byte_off), -1, nullptr, ref.bit_len, ref.bit_pos);
    const NvhPMapping& map = mappings[fr.mapping];
    // ---- floors (Mapping.cs:95-111), as in parse_body ----
    uint32_t energy = 0;
    int err = 0;
    for (int c = 0; c < nch && !err; c++) {
      const int fl = map.chan_floor[c];
      int pc = 0;
      uint16_t* my_posts = posts + ((long long)f * nch + c) * NVH_MAX_POSTS;
      err = decode_floor1<false, false>(T, s_prefix, nullptr, books, floors[fl], p, my_posts, &pc);
      NvhChan cn;
      cn.exec = 0;
      cn.floor = (uint8_t)fl;
      cn.post_count = (uint8_t)pc;
      cn.ov_exec = 0;
      cn.data_off = (uint32_t)(((long long)f * nch + c) * NVH_MAX_POSTS);
      cn.amp = 0.0f;
      ch_out[c] = cn;
      if (pc > 0) energy |= 1u << c;
      pcs |= (unsigned long long)(pc & 0x7F) << (7 * c);
    }
#ifdef NVH_DEBUG
    ft[1] = clock64();
#endif
    if (err || map.submaps != 1) {
      bail = true;
      NVH_WHY(1);
    }
    const bool any_execute = energy != 0;  // computed before ForceEnergy (quirk B-5)
    uint32_t force_e = 0, force_no = 0;
    for (int i = 0; i < map.coupling_steps; i++) {  // Mapping.cs:112-119
      const uint32_t a = 1u << map.coupling_ang[i], m = 1u << map.coupling_mag[i];
      if (((force_e | energy) & ~force_no) & (a | m)) force_e |= a | m;
    }
    for (int j = 0; j < nch; j++)
      if (map.submap_floor[0] != map.chan_floor[j] || map.submap_residue[0] != map.chan_residue[j]) force_no |= 1u << j;
    // ---- the residue pass (Mapping.cs:122-134; Residue0.Decode :119-178) ----
    if (any_execute && !bail) {
      const int residue_idx = map.submap_residue[0];
      const NvhPResidue& r = residues[residue_idx];
      const int r_type = r.type, r_begin = r.begin, r_psize = r.partition_size, r_chs = r.channels, r_rch = r.real_channels,
                r_stages = r.max_stages, r_partvals = r.partvals, cdim = r.class_dims;
      const uint32_t dm_lds = r.decode_map_lds, vis_lds = r.vis_lds;
      int block_size = fr.n;
      if (r_type == 2) block_size *= r_rch;  // Residue2.cs:16-21
      const int end = r.end < block_size / 2 ? r.end : block_size / 2;
      const int n = end - r_begin;
      int partition_count = 0;
      if (n > 0) {
        partition_count = n / r_psize;
        // what the general body tests visit by visit, once: the last partition's longest vector stays inside the buffers
        // (Residue1.cs:12-22, Residue2.cs:27-45), every partition index fits a record, the class rows fit their LDS bytes
        const int last_off = r_begin + (partition_count > 0 ? partition_count - 1 : 0) * r_psize, span = (int)r.span_max;
        const bool inside = r_type == 1 ? last_off + span - 1 < T.block1
                                        : (r_rch > 1 ? (int)__umulhi((uint32_t)last_off, r.rch_magic) + (int)__umulhi((uint32_t)(span - 1), r.rch_magic)
                                                     : last_off + span - 1) < T.block1;
        if (cdim == 0 || r_stages == 0 || (r_type != 1 && r_type != 2) || dm_lds == 0xFFFFFFFFu || vis_lds == 0xFFFFFFFFu || !inside ||
            partition_count > 0xFFFF || r_chs * partition_count > T.cap_parts || span < 1) {
          bail = true;
          NVH_WHY(2);
        }
        if (!bail && partition_count > 0) {
          const NvhPBook class_book = books[r.class_book];
          const bool class_lean = class_book.has_tree && class_book.lds_off != 0xFFFFFFFFu && class_book.prefix_bits >= 1 && class_book.prefix_bits <= 24;
          const uint32_t class_pmask = (1u << (class_book.prefix_bits & 31)) - 1u, class_sdir = sub_words > 1 ? class_book.sub_dir : 0xFFFFFFFFu;
          const int last_base = T.cap_parts;
          auto cls_at = [&](const int i) -> uint8_t& { return s_cls[4 * ((i >> 2) * nt + tid) + (i & 3)]; };
          auto stage_at = [&](const int i) -> uint16_t& { return s_stage16[2 * ((i >> 1) * nt + tid) + (i & 1)]; };
          // Positions -- (partition, channel) in the order the walk visits them: index partition * channels + channel -- have two
          // bytes in LDS each, written when their group's class word is decoded: the class, and the stages in which that class
          // has a book (= how many records its chain has).  A step reads FOUR positions' stage bytes in one word: the first one
          // with a book in this stage is the next visit, the bits in front of it are the records of the chains in front of it.
          // (One position per step, class byte -> book mask -> test, cost 0.6 M of a packet's 1.4 M cycles: 180 steps of two
          // dependent LDS reads, taken by every lane of the wavefront whenever one lane had to.)
          const int npos = partition_count * r_chs, group_pos = cdim * r_chs;
          uint32_t* const s_bmw = reinterpret_cast<uint32_t*>(s_cls) + (size_t)nt * (((size_t)T.cap_parts + 3) / 4);
          auto bm_at = [&](const int i) -> uint8_t& { return reinterpret_cast<uint8_t*>(s_bmw)[4 * ((i >> 2) * nt + tid) + (i & 3)]; };
          s_bmw[((npos - 1) >> 2) * nt + tid] = 0u;  // (the last word's bytes behind the last position)
          uint32_t run = 0, start = 0, run0 = 0;
          int stage = 0, pos = 0, known_end = 0;
          bool going = true;
          while (going) {
            // ---- to the next visit that has a vector to decode ----
#ifdef NVH_DEBUG
            const long long fa0 = clock64();
            ++frounds;
#endif
            unsigned cm = 0;
            for (;;) {
              if (pos >= known_end) {
                if (pos >= npos) {  // next cascade stage (Residue0.cs:132)
                  if (++stage >= r_stages) {
                    going = false;
                    break;
                  }
                  pos = 0; run = 0;
                  continue;
                }
                // stage 0, the class words of the group of partitions that starts here (:137-150), then the chains of its
                // positions in walk order (the tail kernel builds the heads from these rows)
                const int p0 = known_end == 0 ? 0 : (r_chs == 1 ? pos : pos / r_chs);
                for (int cc = 0; cc < r_chs; cc++) {
                  // (the class word the way the entries are decoded -- prefix table, second-level table, whole code inside the
                  // packet --, else the general decode: lanes reach their group boundaries in different steps of a wavefront, so a
                  // class word's cost is paid up to once per lane and group, and the general decode's is ~2 k cycles)
                  int idx = -1;
                  if (class_lean) {
                    const uint32_t node = s_prefix[class_book.lds_off + ((uint32_t)p.buf & class_pmask)], rem = p.total - p.pos;
                    uint32_t val = node >> 8, len = node & 0x7Fu;
                    bool ok = (node & 0x80u) != 0u && len <= rem;
                    if (!ok && (node & 0x80u) == 0u && (node & 0x7Fu) != 0u && (node & 0x7Fu) != 0x7Fu && class_sdir != 0xFFFFFFFFu) {
                      const uint32_t dw = s_sub[class_sdir + (node >> 8)];
                      if (dw != 0u) {
                        const uint32_t e = s_sub[(dw & 0xFFFFFFu) + ((uint32_t)(p.buf >> class_book.prefix_bits) & ((1u << (dw >> 24)) - 1u))];
                        val = e >> 8;
                        len = e & 0x7Fu;
                        ok = (e & 0x80u) != 0u && len <= rem && len <= 32u;
                      }
                    }
                    if (ok) {
                      p.buf >>= len;
                      p.avail -= len;
                      p.pos += len;
                      if (p.avail <= 32u && p.next < p.nwords) {
                        const uint32_t word = p.ahead;
                        p.next++;
                        p.ahead = p.w[p.next];  // (index nwords at most: see br_word)
                        p.buf |= (uint64_t)word << p.avail;
                        p.avail += 32u;
                      }
                      idx = (int)val;
                    }
                  }
                  if (idx < 0) idx = decode_scalar<false, false>(T, s_prefix, nullptr, class_book, p);
                  if (idx < 0 || idx >= r_partvals) {  // the packet ends here, or the reference would fault: not this kernel's
                    bail = true;
                    NVH_WHY(3);
                    break;
                  }
                  for (int d = 0; d < cdim && p0 + d < partition_count; d++) {
                    const uint32_t k = s_prefix[dm_lds + (uint32_t)(idx * cdim + d)];
                    cls_at((p0 + d) * r_chs + cc) = (uint8_t)k;
                    bm_at((p0 + d) * r_chs + cc) = r.book_mask[k];
                  }
                }
                if (bail) {
                  going = false;
                  break;
                }
                known_end = pos + group_pos < npos ? pos + group_pos : npos;
                for (int i = pos; i < known_end; ++i) {
                  const unsigned m = bm_at(i);
                  g_rows[last_base + i] = m ? (int)run0 : -1;
                  run0 += (uint32_t)__popc(m);
                }
              }
              const int k4 = pos & ~3, o = pos & 3, lim = known_end - k4;
              uint32_t w = s_bmw[(pos >> 2) * nt + tid];
              w &= (0xFFFFFFFFu << (8 * o)) & (lim >= 4 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (8 * lim)));
              const uint32_t t = w & (0x01010101u << stage);
              if (t == 0u) {
                run += (uint32_t)__popc(w);
                pos = lim >= 4 ? k4 + 4 : known_end;
                continue;
              }
              const int kk = (__ffs((int)t) - 1) >> 3;
              run += (uint32_t)__popc(w & ~(0xFFFFFFFFu << (8 * kk)));
              start = run;
              cm = (w >> (8 * kk)) & 0xFFu;
              run += (uint32_t)__popc(cm);
              pos = k4 + kk;
              break;
            }
#ifdef NVH_DEBUG
            const long long fa1 = clock64();
            facc[0] += fa1 - fa0;
#endif
            if (!going) break;
            if (run0 > (uint32_t)T.cap_ops) {
              bail = true;
              NVH_WHY(4);
              break;
            }
            // ---- one visit: the entries of a vector (Residue0.cs:157-170) from its descriptor ----
            const int cl = (int)cls_at(pos), c = r_chs == 1 ? 0 : pos % r_chs;
            const uint4 V = *reinterpret_cast<const uint4*>(s_prefix + vis_lds + 4u * (uint32_t)(cl * NVH_MAX_STAGES + stage));
            const int slots = (int)(V.y & 0xFFFFu);
            if (V.x >= NVH_PVIS_SLOW || slots > NVH_PSTG || nent + (uint32_t)slots > (uint32_t)T.cap_ent) {
              bail = true;
              NVH_WHY(5);
              break;
            }
            const uint32_t toff = V.x & 0xFFFFFFu, pmask = (1u << (V.x >> 24)) - 1u;
            // The symbol loop: ONE copy of the consume / store / next-lookup sequence -- a long code only replaces (value, length) in
            // front of it --, the bits left in the packet counted down instead of the position up, the staging address stepped instead
            // of computed: a lone wavefront pays for every instruction and every branch of this loop with its latency.
            int done = 0;
            uint32_t rem = p.total - p.pos;
            uint8_t* sp = reinterpret_cast<uint8_t*>(s_stage16) + 4 * tid;  // entry i: 4 * ((i >> 1) * nt + tid) + 2 * (i & 1) bytes
            const int sp_odd = 4 * nt - 2;
            uint32_t node = s_prefix[toff + ((uint32_t)p.buf & pmask)];
            for (;;) {
              uint32_t val = node >> 8, len = node & 0x7Fu;
              if (!((node & 0x80u) != 0u && len <= rem)) {
                // a code longer than the prefix (Codebook.cs:307-318): its slot's second-level table, else its group of overflow nodes
                // (the buffer is the packet zero-extended, which is what the reference peeks at: DataPacket.cs:168-205)
                const NvhPBook& bk = books[V.z & 0xFFFFu];
                const uint32_t cnt = node & 0x7Fu, ovf = bk.ovf_lds, sdir = sub_words > 1 ? bk.sub_dir : 0xFFFFFFFFu;
                uint32_t hit_len = 0, hit_val = 0;
                if ((node & 0x80u) == 0u && cnt != 0u && cnt != 0x7Fu && sdir != 0xFFFFFFFFu) {
                  const uint32_t dw = s_sub[sdir + (node >> 8)];
                  if (dw != 0u) {
                    const uint32_t e = s_sub[(dw & 0xFFFFFFu) + ((uint32_t)(p.buf >> (V.x >> 24)) & ((1u << (dw >> 24)) - 1u))];
                    if (e & 0x80u) {
                      hit_val = e >> 8;
                      hit_len = e & 0x7Fu;
                    }
                  }
                } else if ((node & 0x80u) == 0u && bk.has_overflow && cnt != 0x7Fu && ovf != 0xFFFFFFFFu) {
                  const uint32_t mb = bk.max_bits;
                  const uint32_t data = (uint32_t)p.buf & (mb >= 32u ? 0xFFFFFFFFu : (1u << mb) - 1u);
                  const uint32_t g = ovf + 2u * (node >> 8);
                  for (uint32_t k = 0; k < cnt; ++k) {
                    const uint32_t bits = s_prefix[g + 2u * k], vl = s_prefix[g + 2u * k + 1u], ln = vl & 0xFFu;
                    if (bits == (data & ((1u << ln) - 1u))) {
                      hit_val = vl >> 8;
                      hit_len = ln;
                      break;
                    }
                  }
                }
                if (hit_len == 0u || hit_len > 32u || hit_len > rem) {  // past the packet's end, a group outside LDS, no such code
                  bail = true;
                  NVH_WHY(6);
                  break;
                }
                val = hit_val;
                len = hit_len;
              }
              p.buf >>= len;
              p.avail -= len;
              rem -= len;
              if (p.avail <= 32u && p.next < p.nwords) {  // (one word restores br_fill's invariant: len <= 32)
                const uint32_t word = p.ahead;
                p.next++;
                p.ahead = p.w[p.next];  // (index nwords at most: see br_word)
                p.buf |= (uint64_t)word << p.avail;
                p.avail += 32u;
              }
              *reinterpret_cast<uint16_t*>(sp) = (uint16_t)val;
              sp += (done & 1) ? sp_odd : 2;
              if (++done >= slots) break;
              node = s_prefix[toff + ((uint32_t)p.buf & pmask)];
            }
            p.pos = p.total - rem;
#ifdef NVH_DEBUG
            const long long fa2 = clock64();
            facc[1] += fa2 - fa1;
#endif
            if (bail) break;
            {
              // the vector's entries to their place: a lone leading one where it starts on an odd entry, then pairs
              uint16_t* __restrict__ eout = entries + ent_base + nent;
              int k = 0;
              if (nent & 1u) {
                eout[0] = stage_at(0);
                k = 1;
              }
              for (; k + 1 < slots; k += 2) *reinterpret_cast<uint32_t*>(eout + k) = (uint32_t)stage_at(k) | ((uint32_t)stage_at(k + 1) << 16);
              if (k < slots) eout[k] = stage_at(k);
            }
            recs[start + ((V.y >> 24) & 7u)] = make_uint2((V.z & 0xFFFF0000u) | nent, V.w | ((uint32_t)c << 25));
            nent += (uint32_t)slots;
            ++nops;
            ++pos;
#ifdef NVH_DEBUG
            facc[2] += clock64() - fa2;
#endif
          }
          nrec_alloc = run0;
          if (run0 > (uint32_t)T.cap_ops) {
            bail = true;
            NVH_WHY(7);
          }
        }
      }
      s_rtype = r_type; s_rch = r_rch; s_psz = r_psize; s_rbegin = r_begin;
      s_parts = partition_count; s_chs = r_chs; s_b1 = r.alias_b1; s_res = residue_idx;
      s_npass = 1;
      npass = 1;
    }
    exec_mask = (force_e | energy) & ~force_no;
    for (int c = 0; c < nch; c++) ch_out[c].exec = (exec_mask >> c) & 1u;
  }
#ifdef NVH_DEBUG
  if (dbg) {
    long long* D = dbg + (long long)f * 24;
    D[20] = bail ? why : 0;
    D[0] = ft[0]; D[1] = ft[0]; D[2] = ft[1]; D[3] = clock64();
    D[8] = facc[1]; D[9] = 0; D[10] = facc[0]; D[11] = frounds; D[21] = facc[2];
  }
#endif
  uint4* const hv = reinterpret_cast<uint4*>(handover + (long long)f * NVH_PHO_WORDS);
  hv[0] = make_uint4(bail ? NVH_PHO_BAIL : 0u, nrec_alloc, nent, nops);
  hv[1] = make_uint4(npass, exec_mask, (uint32_t)pcs, (uint32_t)(pcs >> 32));
  hv[2] = make_uint4((uint32_t)s_rtype, (uint32_t)s_rch, (uint32_t)s_psz, (uint32_t)s_rbegin);
  hv[3] = make_uint4((uint32_t)s_npass, (uint32_t)s_parts, (uint32_t)s_chs | ((uint32_t)s_b1 << 8) | (1u << 16), (uint32_t)s_res);
}


// Second pass: what a frame needs from its neighbours (known only after every lane has parsed its packet): the overlap source's
// execute flags (NvhChan::ov_exec / NvhFrame::ov_exec_mask) -- carry_exec_in: flags of the block carried in from the previous
// batch; the last decoded frame's flags go out for the next one -- and the final word on paired emission (nvh_format.h:
// NVH_EMIT_*).  The host marked the candidates from the geometry alone; a steady-state overlap also needs every channel of both
// blocks to execute (Mapping.cs:104-131 decided that inside k_parse).  A candidate that fails is handed back to k_ola_compact and
// reported (emit_ok = 0: the host then runs k_ola_compact over every frame instead of over its list).
extern "C" __global__ void __launch_bounds__(64)
k_parse_links(int nframes, int channels, NvhFrame* __restrict__ frames, NvhChan* __restrict__ chans,
              const uint32_t* __restrict__ carry_exec_in, uint32_t* __restrict__ carry_exec_out, int last_decoded,
              NvhParseResult* __restrict__ result, uint4* __restrict__ slabs, int stride_vecs) {
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= nframes) return;
  const NvhFrame fr = frames[f];
  uint32_t m = 0;
  if (fr.ov_frame == -2) m = carry_exec_in[0];
  else if (fr.ov_frame >= 0) m = frames[fr.ov_frame].exec_mask;
  if (fr.ov_frame == -2 || fr.ov_frame >= 0) {
    frames[f].ov_exec_mask = m;
    for (int c = 0; c < channels; c++) chans[(long long)f * channels + c].ov_exec = (m >> c) & 1u;
  }
  if (f == last_decoded) carry_exec_out[0] = fr.exec_mask;  // ping-pong with the carried block itself (nvh_api.hip)
  const uint32_t all_ch = channels >= 32 ? 0xFFFFFFFFu : (1u << channels) - 1u;
  auto full = [&](uint32_t x) { return (x & all_ch) == all_ch; };
  uint32_t ef = fr.emit_flags;
  if (ef & (NVH_EMIT_DONE | NVH_EMIT_NEXT)) {
    const bool me = full(fr.exec_mask);
    if (ef & NVH_EMIT_SELF_CARRY) {
      if (!me) ef &= ~(NVH_EMIT_SELF | NVH_EMIT_SELF_CARRY | NVH_EMIT_DONE);
    } else if (ef & NVH_EMIT_DONE) {  // steady: this block over the whole second half of frame f - 1
      if (!(me && f >= 1 && full(frames[f - 1].exec_mask))) ef &= ~(NVH_EMIT_SELF | NVH_EMIT_DONE);
    }
    if (ef & NVH_EMIT_NEXT) {  // frame f + 1 steady over this one
      if (!(me && f + 1 < nframes && full(frames[f + 1].exec_mask))) ef &= ~NVH_EMIT_NEXT;
    }
    if (ef != fr.emit_flags) {
      frames[f].emit_flags = ef;
      atomicAnd(&result->emit_ok, 0);
    }
  }
  // slab mode, mono / stereo: what k_synth_emit needs to know about the overlaps this frame emits goes into its slab's header
  // (k_parse left those fields clear; host_slab.cpp writes the same for host-parsed batches)
  if (slabs && channels <= 2 && fr.n != 0 && (ef & (NVH_EMIT_SELF | NVH_EMIT_NEXT | NVH_EMIT_CARRY_OUT | NVH_EMIT_DONE))) {
    uint32_t* H = reinterpret_cast<uint32_t*>(slabs + (long long)f * stride_vecs);
    uint32_t w0 = H[0];
    if (ef & NVH_EMIT_CARRY_OUT) {
      w0 |= (uint32_t)NVH_SLABX_CARRY_OUT << 16;
      H[8 + 2] = fr.window_off;
    }
    if (ef & NVH_EMIT_SELF_CARRY) w0 |= (uint32_t)NVH_SLABX_SELF_CARRY << 16;
    if (ef & NVH_EMIT_DONE) w0 |= (uint32_t)NVH_SLABX_DONE << 16;
    if (ef & (NVH_EMIT_SELF | NVH_EMIT_NEXT | NVH_EMIT_DONE)) {
      H[8 + 2] = fr.window_off; H[8 + 3] = fr.ov_window_off; H[8 + 6] = (uint32_t)fr.out_pos;
      const bool nxt = (ef & NVH_EMIT_NEXT) && f + 1 < nframes;
      H[8 + 5] = NVH_SLAB_GEO(fr.ov_n, nxt ? frames[f + 1].n : 0, fr.start, fr.valid);
      if (ef & NVH_EMIT_SELF) w0 |= (uint32_t)NVH_SLAB_EMIT_SELF << 24;
      if (nxt) {
        const NvhFrame nx = frames[f + 1];
        H[8 + 4] = nx.window_off; H[8 + 7] = (uint32_t)nx.out_pos;
        w0 |= (uint32_t)NVH_SLAB_EMIT_NEXT << 24;
      }
    }
    H[0] = w0;
  }
}

// The batch's result words for the host: stored into page-locked host memory by the device itself, so that nothing the
// pipelined path waits for is a copy command (see k_parse_fetch below).
extern "C" __global__ void k_parse_result_out(const NvhParseResult* __restrict__ dev, NvhParseResult* __restrict__ host) {
  const unsigned i = threadIdx.x;
  if (i < sizeof(NvhParseResult) / 4) __builtin_nontemporal_store(((const uint32_t*)dev)[i], (uint32_t*)host + i);
}

// A GPU-parse batch's host-written input, fetched by the device itself from page-locked host memory: the staging block
// (frames, channel records, packet references, left-over list) and the packet pool, plus the eight zero bytes behind
// the pool and the result block's initial state.  Copy commands for the same bytes ran at a tenth of the link rate
// while the previous batch's PCM read-back was in flight (3.3 ms instead of 0.36 ms for 24 MB); the loads of a kernel
// do not care.
extern "C" __global__ void __launch_bounds__(256)
k_parse_fetch(const uint4* __restrict__ stage_h, uint4* __restrict__ stage_d, long long stage_n16,
              const uint8_t* __restrict__ pool_h, uint8_t* __restrict__ pool_d, long long pool_bytes,
              NvhParseResult* __restrict__ result) {
  const long long step = (long long)gridDim.x * 256;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  for (long long i = t; i < stage_n16; i += step) stage_d[i] = stage_h[i];
  if (pool_d) {
    const long long n16 = pool_bytes >> 4;
    const uint4* ph = (const uint4*)pool_h;  // (both pool addresses are 256-byte aligned)
    uint4* pd = (uint4*)pool_d;
    for (long long i = t; i < n16; i += step) pd[i] = ph[i];
    if (t < 24) {  // the last partial vector, then the zero bytes a reader running off a packet's end sees
      const long long at = (n16 << 4) + t;
      if (at < pool_bytes) pool_d[at] = pool_h[at];
      else if (at < pool_bytes + 8) pool_d[at] = 0;
    }
  }
  if (t == 0) {
    NvhParseResult init{};
    init.err_frame = 0x7FFFFFFF;
    init.links_ok = 1;
    init.emit_ok = 1;
    *result = init;
  }
}
