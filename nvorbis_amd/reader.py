"""Host-side mirror of NVorbis' reader surface over the C ABI.

`VorbisReader.ReadSamples` / `StreamDecoder.Read` keep the reference's semantics
(VorbisReader.cs:336-363, StreamDecoder.cs:320-389): counts are trimmed to a multiple of the channel
count, partial reads are allowed, 0 is returned at the end of the stream.  Internally packets are
parsed a batch ahead on the host and synthesised on the GPU one batch per launch sequence; a PCM ring
is drained by the reads.  Python stands in for the C# shim of INTEGRATION.md (no .NET in this image).
"""
import ctypes as C
import threading

import numpy as np

from . import native
from .native import check, lib


class PacketArray:
    """Packets of one logical stream in one contiguous buffer (what nvh_ogg_demux produces): packet i is
    bytes[offsets[i]:offsets[i+1]].  Feeds nvh_stream_push_packets without one FFI call per packet."""

    def __init__(self, data, offsets, granules, flags):
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = self.offsets.size - 1
        self.granules = np.ascontiguousarray(granules[:n], dtype=np.int64) if n else np.zeros(1, np.int64)
        self.flags = np.ascontiguousarray(flags[:n], dtype=np.uint8) if n else np.zeros(1, np.uint8)
        self._n = n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if i < 0 or i >= self._n:
            raise IndexError(i)
        return self.data[self.offsets[i]:self.offsets[i + 1]].tobytes()

    @staticmethod
    def from_list(packets, granules=None, flags=None):
        """The same from a list of packet byte strings."""
        if isinstance(packets, PacketArray):
            return packets
        n = len(packets)
        offs = np.zeros(n + 1, np.int64)
        offs[1:] = np.cumsum([len(p) for p in packets])
        blob = np.frombuffer(b"".join(packets) or b"\0", dtype=np.uint8)
        gr = np.asarray(granules if granules is not None else [-1] * n, np.int64)
        fl = np.asarray(flags if flags is not None else [0] * n, np.uint8)
        return PacketArray(blob, offs, gr if n else np.zeros(1, np.int64), fl if n else np.zeros(1, np.uint8))


def ogg_stream_count(data: bytes):
    """Number of logical streams in an Ogg file (multiplexed or chained), Vorbis or not."""
    n, total, ns = C.c_int(0), C.c_int64(0), C.c_int(0)
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    check(lib().nvh_ogg_demux_stream(buf, len(data), 0, None, 0, None, None, None, 0, C.byref(n), C.byref(total), C.byref(ns)),
          "nvh_ogg_demux_stream")
    return ns.value


def demux_ogg_array(data: bytes, stream_index=0, forward_only=False):
    """Logical stream `stream_index` (default: the first) of an Ogg file as a PacketArray.

    Delivers packets the way NVorbis' seekable reader does (Ogg/PacketProvider.cs:324-438), or -- forward_only -- the way its
    reader for sources that cannot seek does (Ogg/ForwardOnlyPacketProvider.cs:119-246)."""
    L = lib()
    fn = L.nvh_ogg_demux_forward if forward_only else L.nvh_ogg_demux_stream
    n = C.c_int(0)
    total = C.c_int64(0)
    k = int(stream_index)
    # the file's bytes where they lie (no copy under the GIL: a worker pool of demuxing threads would take turns at it)
    src = np.frombuffer(data if data else b"\0", dtype=np.uint8)
    buf = C.c_void_p(src.ctypes.data)
    # One call where a guess at the packet count holds (a packet every 48 bytes or fewer is not audio): the packets of a stream
    # cannot be longer than its file, so the byte buffer is sized by the file.  The library then demultiplexes once and copies
    # once; the sizing call + fill call below cost a fingerprint of the file each on top.
    cap_n = len(data) // 48 + 64
    if len(data) >= 4096:
        pk = np.empty(len(data), dtype=np.uint8)
        offs = np.empty(cap_n + 1, dtype=np.int64)
        gran = np.empty(cap_n, dtype=np.int64)
        flags = np.empty(cap_n, dtype=np.uint8)
        rc = fn(buf, len(data), k, pk.ctypes.data, pk.size, offs.ctypes.data, gran.ctypes.data, flags.ctypes.data, cap_n,
                C.byref(n), C.byref(total), None)
        if rc == native.OK:
            m = n.value
            return PacketArray(pk[:max(total.value, 1)], offs[:m + 1].copy(), gran[:max(m, 1)].copy(), flags[:max(m, 1)].copy())
        if rc != native.ERR_ARGUMENT:
            check(rc, "nvh_ogg_demux")
    check(fn(buf, len(data), k, None, 0, None, None, None, 0, C.byref(n), C.byref(total), None), "nvh_ogg_demux")
    pk = np.zeros(max(total.value, 1), dtype=np.uint8)
    offs = np.zeros(n.value + 1, dtype=np.int64)
    gran = np.zeros(max(n.value, 1), dtype=np.int64)
    flags = np.zeros(max(n.value, 1), dtype=np.uint8)
    check(fn(buf, len(data), k, pk.ctypes.data, pk.size, offs.ctypes.data, gran.ctypes.data,
             flags.ctypes.data, n.value, C.byref(n), C.byref(total), None), "nvh_ogg_demux")
    return PacketArray(pk, offs[:n.value + 1], gran, flags)


_index_scratch = threading.local()  # index_ogg_array: the calling thread's output arrays


def index_ogg_array(data: bytes, stream_index=0):
    """The index form of demux_ogg_array (nvh_ogg_index_packets): the stream's packet list without page checksums and packet
    bodies -- the three headers whole, every audio packet as its first (up to) 8 bytes -- as (PacketArray, payload_bytes); what a
    sizing pass hands to Stream.index_packets.  A later demux_ogg_array of the same bytes must find the same packet count and
    the same payload size (PacketArray.data.size), else a page was damaged and the index does not hold."""
    L = lib()
    n = C.c_int(0)
    total = C.c_int64(0)
    payload = C.c_int64(0)
    src = np.frombuffer(data if data else b"\0", dtype=np.uint8)
    # heads + headers: 8 bytes per packet and the setup header; a page (>= 27 bytes of header) starts at most 255 packets, and only
    # lacing values of 0..254 end one, so len / 27 + len / 64 is far beyond any audio stream -- the call says so if it is not
    cap_n = len(data) // 48 + 64
    cap_b = min(len(data), 8 * cap_n + (1 << 16)) + 64
    # The call writes into scratch arrays of the calling thread, kept from file to file (a pool of index threads would otherwise
    # allocate and release four arrays of up to a megabyte per file side by side: the pass then does not scale with its threads --
    # 502 corpus files: 0.22 s on one thread, 0.22 s on eight; with the scratch 0.04 s on eight); what the call filled is copied out.  (The scratch lives as long as its thread: at most the
    # largest index it has produced, ~0.2 % of that file's size.)
    sc = getattr(_index_scratch, "arrays", None)
    for _ in range(2):
        if sc is None or sc[0].size < cap_b or sc[2].size < cap_n:
            grow_n = max(cap_n, 2 * sc[2].size if sc is not None else 0)
            grow_b = max(cap_b, 2 * sc[0].size if sc is not None else 0)
            sc = (np.empty(grow_b, dtype=np.uint8), np.empty(grow_n + 1, dtype=np.int64), np.empty(grow_n, dtype=np.int64),
                  np.empty(grow_n, dtype=np.uint8))
            _index_scratch.arrays = sc
        pk, offs, gran, flags = sc
        rc = L.nvh_ogg_index_packets(C.c_void_p(src.ctypes.data), len(data), int(stream_index), pk.ctypes.data, pk.size, offs.ctypes.data,
                                     gran.ctypes.data, flags.ctypes.data, gran.size, C.byref(n), C.byref(total), C.byref(payload), None)
        if rc == native.ERR_ARGUMENT and (n.value > gran.size or total.value > pk.size):
            cap_n, cap_b = n.value + 1, total.value + 64
            continue
        check(rc, "nvh_ogg_index_packets")
        break
    m = n.value
    return PacketArray(pk[:max(total.value, 1)].copy(), offs[:m + 1].copy(), gran[:max(m, 1)].copy(), flags[:max(m, 1)].copy()), int(payload.value)


def demux_ogg(data: bytes, forward_only=False):
    """First logical stream of an Ogg file -> (list of packet bytes, granules, flags)."""
    pa = demux_ogg_array(data, 0, forward_only)
    n = len(pa)
    return [pa[i] for i in range(n)], pa.granules[:n].copy(), pa.flags[:n].copy()


class Comm:
    """nvh_comm: the corpus gather through RCCL's C API (include/nvorbis_hip.h, "multi-GPU"; nvh_comm.hip) -- what a host
    without torch.distributed calls.  One per process, on the process's Context.  `id_bytes`: Comm.unique_id() of rank 0,
    handed to the other ranks by the caller."""

    SELF_P2P = 1  # NVH_GATHER_SELF_P2P

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(lib().nvh_comm_unique_id(buf), "nvh_comm_unique_id")
        return bytes(buf)

    def __init__(self, ctx, id_bytes, rank, world):
        if len(id_bytes) != 128:
            raise ValueError("the id is the 128 bytes Comm.unique_id() returned")
        self._h = C.c_void_p()
        self.rank, self.world = int(rank), int(world)
        self._ctx = ctx  # (the context must outlive the communicator)
        buf = (C.c_uint8 * 128).from_buffer_copy(id_bytes)
        check(lib().nvh_comm_create(ctx._h, buf, int(rank), int(world), C.byref(self._h)), "nvh_comm_create")

    def allgather_i64(self, mine):
        """`mine`: a list of n ints; returns world lists of n ints (rank-major)."""
        n = len(mine)
        src = (C.c_int64 * n)(*[int(v) for v in mine])
        out = (C.c_int64 * (self.world * n))()
        check(lib().nvh_comm_allgather_i64(self._h, src, n, out), "nvh_comm_allgather_i64")
        return [[int(out[r * n + k]) for k in range(n)] for r in range(self.world)]

    def gather_pcm(self, d_send, send_count, d_recv, counts, root=0, flags=0):
        """d_send / d_recv: device addresses (ints); counts[world]: every rank's total number of floats."""
        arr = (C.c_int64 * self.world)(*[int(v) for v in counts])
        check(lib().nvh_comm_gather_pcm(self._h, C.c_void_p(d_send), int(send_count), C.c_void_p(d_recv), arr, int(root), int(flags)),
              "nvh_comm_gather_pcm")

    def close(self):
        if self._h:
            lib().nvh_comm_destroy(self._h)
            self._h = C.c_void_p()


class Context:
    """nvh_ctx: one GPU + one HIP stream."""

    def __init__(self, device=0, hip_stream=None):
        self._h = C.c_void_p()
        self.device = int(device)
        check(lib().nvh_ctx_create(int(device), C.byref(self._h)), "nvh_ctx_create")
        if hip_stream is not None:
            self.set_hip_stream(hip_stream)

    def set_hip_stream(self, hip_stream):
        check(lib().nvh_ctx_set_hip_stream(self._h, C.c_void_p(hip_stream)), "nvh_ctx_set_hip_stream")

    def synchronize(self):
        check(lib().nvh_ctx_synchronize(self._h), "nvh_ctx_synchronize")

    def set_parse_lanes(self, lanes):
        """Packets per wavefront of the GPU packet parser for this context's streams (0 = automatic); see the header."""
        check(lib().nvh_ctx_set_parse_lanes(self._h, int(lanes)), "nvh_ctx_set_parse_lanes")

    def mdct_reverse(self, n, batch, d_ptr, stride):
        """IMdct.Reverse on `batch` device buffers (Contracts/IMdct.cs:5)."""
        check(lib().nvh_mdct_reverse(self._h, int(n), int(batch), C.c_void_p(d_ptr), int(stride)), "nvh_mdct_reverse")

    def overlap_buffers(self, d_previous, d_next, prev_start, prev_stop, next_start, channels, plane_stride):
        """StreamDecoder.OverlapBuffers on device planes [channels][plane_stride]."""
        check(lib().nvh_overlap_buffers(self._h, C.c_void_p(d_previous), C.c_void_p(d_next), int(prev_start), int(prev_stop),
                                        int(next_start), int(channels), int(plane_stride)), "nvh_overlap_buffers")

    def copy_buffer(self, d_planes, start, count, channels, plane_stride, d_target, clip=True):
        """ClippingCopyBuffer / CopyBuffer: planar device planes -> interleaved device target; returns HasClipped."""
        clipped = C.c_int(0)
        check(lib().nvh_copy_buffer(self._h, C.c_void_p(d_planes), int(start), int(count), int(channels), int(plane_stride),
                                    C.c_void_p(d_target), 1 if clip else 0, C.byref(clipped)), "nvh_copy_buffer")
        return bool(clipped.value)

    def inverse_couple(self, d_magnitude, d_angle, count):
        """One inverse square-polar coupling step over two device vectors, in place (Mapping.cs:150-178)."""
        check(lib().nvh_inverse_couple(self._h, C.c_void_p(d_magnitude), C.c_void_p(d_angle), int(count)), "nvh_inverse_couple")

    def close(self):
        if self._h:
            lib().nvh_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """nvh_batch: a parsed batch of frames resident in HBM (descriptors + work planes)."""

    def __init__(self, handle, stream):
        self._h = handle
        self._stream = stream  # keeps the owning nvh_stream (and its nvh_ctx) alive until the batch is freed
        self.channels = stream.channels
        fr, cf = C.c_int(0), C.c_int(0)
        smp, db = C.c_int64(0), C.c_int64(0)
        check(lib().nvh_batch_info(self._h, C.byref(fr), C.byref(cf), C.byref(smp), C.byref(db)), "nvh_batch_info")
        self.frames, self.chan_frames, self.samples, self.descriptor_bytes = fr.value, cf.value, smp.value, db.value

    def stats(self):
        out = (C.c_int64 * 8)()
        check(lib().nvh_batch_stats(self._h, out), "nvh_batch_stats")
        keys = ["frames", "chan_frames", "passes", "ops", "entries", "posts", "coeffs", "reserved"]
        return {k: out[i] for i, k in enumerate(keys)}

    def kernels(self):
        """Kernel names behind the four timing slots of the last launch ("-" = empty slot)."""
        buf = C.create_string_buffer(256)
        check(lib().nvh_batch_kernels(self._h, buf, 256), "nvh_batch_kernels")
        return buf.value.decode().split(",")

    def synth(self, d_pcm_ptr, capacity):
        check(lib().nvh_batch_synth(self._h, C.c_void_p(d_pcm_ptr), int(capacity)), "nvh_batch_synth")

    def time(self, d_pcm_ptr, capacity, iters, per_kernel=True):
        total = C.c_float(0)
        km = (C.c_float * 4)()
        check(lib().nvh_batch_time(self._h, C.c_void_p(d_pcm_ptr), int(capacity), int(iters), C.byref(total),
                                   km if per_kernel else None), "nvh_batch_time")
        return total.value, [km[i] for i in range(4)]

    def free(self):
        if self._h:
            lib().nvh_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    """nvh_stream: setup tables in HBM + host parser + overlap state.  ctx=None -> host-only (parse, no synthesis)."""

    def __init__(self, ctx, id_pkt, comment_pkt, setup_pkt):
        self._ctx = ctx
        self._h = C.c_void_p()
        check(lib().nvh_stream_open(ctx._h if ctx is not None else None, id_pkt, len(id_pkt), comment_pkt,
                                    len(comment_pkt) if comment_pkt is not None else 0, setup_pkt, len(setup_pkt),
                                    C.byref(self._h)), "nvh_stream_open")
        ch, sr, b0, b1 = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().nvh_stream_info(self._h, C.byref(ch), C.byref(sr), C.byref(b0), C.byref(b1)), "nvh_stream_info")
        self.channels, self.sample_rate, self.block0, self.block1 = ch.value, sr.value, b0.value, b1.value
        self.parse_errors = []

    def bitrates(self):
        """(UpperBitrate, NominalBitrate, LowerBitrate) of the identification header (StreamDecoder.cs:191-199)."""
        u, n, l = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().nvh_stream_bitrates(self._h, C.byref(u), C.byref(n), C.byref(l)), "nvh_stream_bitrates")
        return u.value, n.value, l.value

    def push_packet(self, data, granule=-1, flags=0):
        check(lib().nvh_stream_push_packet(self._h, data, len(data), int(granule), int(flags)), "nvh_stream_push_packet")

    def push_packets(self, pa, first, max_packets):
        """Push packets pa[first:] (a PacketArray) until max_packets were taken or the stream saw its EOS packet;
        returns the number consumed."""
        took = C.c_int(0)
        n = len(pa) - first
        rc = lib().nvh_stream_push_packets(self._h, pa.data.ctypes.data, pa.offsets[first:].ctypes.data,
                                           pa.granules[first:].ctypes.data, pa.flags[first:].ctypes.data, int(n),
                                           int(max_packets), C.byref(took))
        if rc != native.OK:
            # packets [first, first + took) were consumed, packet first + took is the one the reference throws on; it is
            # consumed too (StreamDecoder.cs:465-530: `packet?.Done()` runs in the finally block).  The caller gets the
            # count through the exception so that it does not push them again.
            e = native.NvhError(rc, "nvh_stream_push_packets")
            e.consumed = took.value + 1
            raise e
        return took.value

    def push_end(self):
        check(lib().nvh_stream_push_end(self._h), "nvh_stream_push_end")

    def pending(self):
        fr, smp = C.c_int(0), C.c_int64(0)
        check(lib().nvh_stream_pending(self._h, C.byref(fr), C.byref(smp)), "nvh_stream_pending")
        return fr.value, smp.value

    def pending_geometry(self):
        fr, _ = self.pending()
        out = np.zeros((max(fr, 1), 8), dtype=np.int32)
        check(lib().nvh_stream_pending_geometry(self._h, out.ctypes.data, max(fr, 1)), "nvh_stream_pending_geometry")
        return out[:fr]

    def kernels(self):
        """Kernel names behind the four timing slots of the stream's last synthesis launch ("-" = empty slot)."""
        buf = C.create_string_buffer(256)
        check(lib().nvh_stream_kernels(self._h, buf, 256), "nvh_stream_kernels")
        return buf.value.decode().split(",")

    def pending_slabs(self):
        """The pending frames as the synthesis kernels fetch them (per-frame slabs, nvh_format.h: NvhSlabHdr), written on the
        host: (uint32 words of all slabs back to back, first 16-byte unit of every frame's slab [frames + 1])."""
        fr, _ = self.pending()
        need = C.c_int64(0)
        first = np.zeros(fr + 1, dtype=np.uint32)
        rc = lib().nvh_stream_pending_slabs(self._h, None, 0, C.byref(need), first.ctypes.data, fr + 1)
        if rc not in (0, -2):
            check(rc, "nvh_stream_pending_slabs")
        buf = np.zeros(max(need.value // 4, 4), dtype=np.uint32)
        check(lib().nvh_stream_pending_slabs(self._h, buf.ctypes.data, buf.nbytes, C.byref(need), first.ctypes.data, fr + 1),
              "nvh_stream_pending_slabs")
        return buf[:need.value // 4], first

    def lattice_pool(self):
        """The setup's lattice pool (uint32 words: per lattice codebook its component values as float bits, then its power
        reciprocals): what the lattice offset of a slab record points into."""
        need = C.c_int64(0)
        lib().nvh_stream_lattice_pool(self._h, None, 0, C.byref(need))
        buf = np.zeros(max(int(need.value), 1), np.uint32)
        check(lib().nvh_stream_lattice_pool(self._h, buf.ctypes.data, buf.size, C.byref(need)), "nvh_stream_lattice_pool")
        return buf[:int(need.value)]

    def vq_pool(self):
        """The setup's VQ table pool (float32): what the lattice-pool word of a book with an explicit table points into."""
        need = C.c_int64(0)
        lib().nvh_stream_vq_pool(self._h, None, 0, C.byref(need))
        buf = np.zeros(max(int(need.value), 1), np.float32)
        check(lib().nvh_stream_vq_pool(self._h, buf.ctypes.data, buf.size, C.byref(need)), "nvh_stream_vq_pool")
        return buf[:int(need.value)]

    def position(self):
        pos, em, eos = C.c_int64(0), C.c_int64(0), C.c_int(0)
        check(lib().nvh_stream_position(self._h, C.byref(pos), C.byref(em), C.byref(eos)), "nvh_stream_position")
        return pos.value, em.value, bool(eos.value)

    def position_state(self):
        """(_hasPosition, _currentPosition) after everything pushed so far was read."""
        h, p = C.c_int(0), C.c_int64(0)
        check(lib().nvh_stream_position_state(self._h, C.byref(h), C.byref(p)), "nvh_stream_position_state")
        return bool(h.value), p.value

    def set_position_state(self, has_position, position):
        check(lib().nvh_stream_set_position_state(self._h, 1 if has_position else 0, int(position)), "nvh_stream_set_position_state")

    def index_packets(self, pa, first=3):
        """Integer geometry of the audio packets pa[first:] as a serial decoder sees them (nvh_stream_index_packets).
        Returns (position_after, emitted_after, state_after, total_emitted); state bits: 1 decodes, 2 safe lead-in,
        4 _hasPosition, 8 _eosFound."""
        n = max(0, len(pa) - first)
        pos, em, st = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.uint8)
        total = C.c_int64(0)
        check(lib().nvh_stream_index_packets(self._h, pa.data.ctypes.data, pa.offsets[first:].ctypes.data if n else pa.offsets.ctypes.data,
                                             pa.granules[first:].ctypes.data if n else None, pa.flags[first:].ctypes.data if n else None,
                                             n, pos.ctypes.data, em.ctypes.data, st.ctypes.data, C.byref(total)), "nvh_stream_index_packets")
        return pos[:n], em[:n], st[:n], total.value

    def index_total(self, pa, first=3):
        """total_emitted of index_packets alone (the per-packet outputs of nvh_stream_index_packets may be NULL): what a sizing pass
        asks, without three arrays per file."""
        n = max(0, len(pa) - first)
        total = C.c_int64(0)
        check(lib().nvh_stream_index_packets(self._h, pa.data.ctypes.data, pa.offsets[first:].ctypes.data if n else pa.offsets.ctypes.data,
                                             pa.granules[first:].ctypes.data if n else None, pa.flags[first:].ctypes.data if n else None,
                                             n, None, None, None, C.byref(total)), "nvh_stream_index_packets")
        return total.value

    def packet_sample_count(self, packet, is_resync=False):
        """StreamDecoder.GetPacketGranules (StreamDecoder.cs:630-647)."""
        n = C.c_int(0)
        check(lib().nvh_stream_packet_sample_count(self._h, packet, len(packet), 1 if is_resync else 0, C.byref(n)),
              "nvh_stream_packet_sample_count")
        return n.value

    def reset(self):
        """ResetDecoder (StreamDecoder.cs:295-305)."""
        check(lib().nvh_stream_reset(self._h), "nvh_stream_reset")
        # nvh_stream_reset synchronises and abandons the outstanding flights (both native slot indices go back to 0):
        # the Python side of the pairing follows, or the next begin / end would address different buffers
        self._pipe_next = self._pipe_first = self._pipe_out = 0

    def drop_pending(self):
        check(lib().nvh_stream_drop_pending(self._h), "nvh_stream_drop_pending")

    def mode_decode(self, packet, d_block):
        """IMode.Decode on one packet: the windowed block before overlap into d_block [channels][block1] (device pointer).
        Returns None if the packet is not decoded, else (block_size, start, valid, total)."""
        dec, bs, a, b, c = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().nvh_mode_decode(self._h, packet, len(packet), C.c_void_p(d_block), C.byref(dec), C.byref(bs), C.byref(a),
                                    C.byref(b), C.byref(c)), "nvh_mode_decode")
        return (bs.value, a.value, b.value, c.value) if dec.value else None

    def residue_decode(self, residue_index, packet, bit_offset, block_size, d_buffer, any_channel_decodes=True):
        """IResidue.Decode: adds the vectors packet[bit_offset:] encodes into the device planes [channels][block1] at
        d_buffer; returns the number of bits consumed."""
        bits = C.c_int(0)
        check(lib().nvh_residue_decode(self._h, int(residue_index), packet, len(packet), int(bit_offset),
                                       1 if any_channel_decodes else 0, int(block_size), C.c_void_p(d_buffer), C.byref(bits)),
              "nvh_residue_decode")
        return bits.value

    def window_apply(self, mode_index, prev_flag, next_flag, batch, d_buf, stride):
        """Mode.Decode's window loop on `batch` device buffers."""
        check(lib().nvh_window_apply(self._h, int(mode_index), int(prev_flag), int(next_flag), int(batch), C.c_void_p(d_buf),
                                     int(stride)), "nvh_window_apply")

    def mode_info(self, mode_index):
        """(block flag, block size, mapping) of one of the stream's modes; None past the last one."""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        if lib().nvh_stream_mode_info(self._h, int(mode_index), C.byref(a), C.byref(b), C.byref(c)) != 0:
            return None
        return bool(a.value), b.value, c.value

    def floor_info(self, floor_index):
        """(type, post count, range) of one of the stream's floors."""
        t, pc, rg = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().nvh_stream_floor_info(self._h, int(floor_index), C.byref(t), C.byref(pc), C.byref(rg)), "nvh_stream_floor_info")
        return t.value, pc.value, rg.value

    def floor0_apply(self, floor_index, block_size, amps, coeffs, d_residue, stride):
        """IFloor.Apply (Floor0.cs:152-212) on a batch of device vectors: amps [batch], coeffs [batch][>= order]."""
        amps = np.ascontiguousarray(amps, dtype=np.float32)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.float32)
        batch = amps.shape[0]
        status = np.zeros(batch, np.int32)
        check(lib().nvh_floor0_apply(self._h, int(floor_index), int(block_size), batch, amps.ctypes.data, coeffs.ctypes.data,
                                     int(coeffs.shape[1]), C.c_void_p(d_residue), int(stride), status.ctypes.data), "nvh_floor0_apply")
        return status

    def floor1_apply(self, floor_index, block_size, posts, post_counts, d_residue, stride):
        """IFloor.Apply (Floor1.cs:186-341) on a batch of device vectors: posts [batch][64] raw Unpack values, post_counts
        [batch] (0 or the floor's post count), d_residue a device pointer to [batch][stride] floats of which the first
        block_size/2 of each row are scaled (or cleared).  Returns the per-item status array."""
        posts = np.ascontiguousarray(posts, dtype=np.int32)
        post_counts = np.ascontiguousarray(post_counts, dtype=np.int32)
        batch = post_counts.shape[0]
        if posts.shape != (batch, 64):
            raise ValueError("posts must be [batch][64]")
        status = np.zeros(batch, np.int32)
        check(lib().nvh_floor1_apply(self._h, int(floor_index), int(block_size), batch, posts.ctypes.data, post_counts.ctypes.data,
                                     C.c_void_p(d_residue), int(stride), status.ctypes.data), "nvh_floor1_apply")
        return status

    def set_gpu_parse(self, on):
        """Parse packets on the GPU (kernels_parse.hip); raises NvhError(UNSUPPORTED) for ineligible stream shapes."""
        check(lib().nvh_stream_set_gpu_parse(self._h, 1 if on else 0), "nvh_stream_set_gpu_parse")

    def set_clip(self, on):
        check(lib().nvh_stream_set_clip(self._h, 1 if on else 0), "nvh_stream_set_clip")

    def has_clipped(self):
        v = C.c_int(0)
        check(lib().nvh_stream_has_clipped(self._h, C.byref(v)), "nvh_stream_has_clipped")
        return bool(v.value)

    def synth_host(self, pinned=False, out=None):
        """Synthesise the pending batch; returns interleaved float32 PCM (numpy).

        out: a contiguous float32 array to write into (must hold the batch); the written prefix is returned.

        pinned=True: the result is a view of a page-locked buffer owned by this stream (written by the copy engine
        directly, no extra copy) and stays valid until the next call."""
        _, smp = self.pending()
        n = max(smp * self.channels, 1)
        wr = C.c_int64(0)
        if out is not None:
            if out.dtype != np.float32 or not out.flags["C_CONTIGUOUS"] or out.size < smp * self.channels:
                raise ValueError("out must be a contiguous float32 array that holds the pending batch")
            if out.size == 0:
                out = np.empty(1, dtype=np.float32)
            n = out.size
        elif pinned:
            if getattr(self, "_pin_cap", 0) < n:
                if getattr(self, "_pin_ptr", None):
                    lib().nvh_pinned_free(self._pin_ptr)
                p = C.c_void_p()
                cap = max(n, 2 * getattr(self, "_pin_cap", 0))
                check(lib().nvh_pinned_alloc(cap * 4, C.byref(p)), "nvh_pinned_alloc")
                self._pin_ptr, self._pin_cap = p, cap
                self._pin_arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(cap,))
            out = self._pin_arr
        else:
            out = np.empty(n, dtype=np.float32)
        rc = lib().nvh_stream_synth(self._h, out.ctypes.data, None, n, C.byref(wr))
        self._note_parse_error(rc, wr.value, "nvh_stream_synth")
        return out[:wr.value]

    # ---- pipelined read-back (nvh_stream_synth_begin / _end) ----
    def _pipe_buffer(self, k, n):
        bufs = getattr(self, "_pipe", None)
        if bufs is None:
            bufs = self._pipe = [[None, 0, None], [None, 0, None]]  # [pointer, capacity in floats, numpy view]
        ptr, cap, arr = bufs[k]
        if cap < n:
            if ptr:
                lib().nvh_pinned_free(ptr)
            p = C.c_void_p()
            cap = max(n, 2 * cap)
            check(lib().nvh_pinned_alloc(cap * 4, C.byref(p)), "nvh_pinned_alloc")
            bufs[k] = [p, cap, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(cap,))]
        return bufs[k][2]

    def synth_begin(self):
        """Queue the pending batch (upload, GPU parse, synthesis, transfer of the PCM on a copy stream) and return at once.
        Two batches may be outstanding; synth_end() hands them back in order."""
        if getattr(self, "_pipe_out", 0) >= 2:
            # refuse before touching a buffer: slot k is still the DMA destination of the oldest outstanding batch
            raise native.NvhError(native.ERR_ARGUMENT, "nvh_stream_synth_begin (two batches are outstanding: call synth_end first)")
        _, smp = self.pending()
        n = max(smp * self.channels, 1)
        k = getattr(self, "_pipe_next", 0)
        out = self._pipe_buffer(k, n)
        exp = C.c_int64(0)
        check(lib().nvh_stream_synth_begin(self._h, out.ctypes.data, out.size, C.byref(exp)), "nvh_stream_synth_begin")
        # only a begin that succeeded occupies a slot
        self._pipe_next = k ^ 1
        self._pipe_out = getattr(self, "_pipe_out", 0) + 1
        return exp.value

    def synth_end(self):
        """PCM of the oldest outstanding batch: a view of a page-locked buffer that stays valid until the begin after next."""
        if getattr(self, "_pipe_out", 0) <= 0:
            raise native.NvhError(native.ERR_ARGUMENT, "nvh_stream_synth_end (nothing is outstanding)")
        k = getattr(self, "_pipe_first", 0)
        wr = C.c_int64(0)
        rc = lib().nvh_stream_synth_end(self._h, C.byref(wr))
        if rc in (native.ERR_ARGUMENT, native.ERR_DEVICE, native.ERR_NO_GPU):
            # the native side returned before it retired the flight (nothing outstanding / the wait itself failed):
            # the pairing is unchanged
            raise native.NvhError(rc, "nvh_stream_synth_end")
        # any other outcome has retired the native flight (nvh_api.hip pops the slot before it reports a runtime or parse error)
        self._pipe_first = k ^ 1
        self._pipe_out -= 1
        self._note_parse_error(rc, wr.value, "nvh_stream_synth_end")
        return self._pipe[k][2][:wr.value]

    def synth_device(self, d_ptr, capacity):
        wr = C.c_int64(0)
        rc = lib().nvh_stream_synth(self._h, None, C.c_void_p(d_ptr), int(capacity), C.byref(wr))
        self._note_parse_error(rc, wr.value, "nvh_stream_synth")
        return wr.value

    def _note_parse_error(self, rc, written, where):
        """A synthesis call that returns an error code together with PCM (GPU-parse mode: a packet of the batch made the
        parser fail and the batch was parsed again without them): the PCM is complete; the errors are kept in
        `parse_errors` = [(NvhError, floats of this batch's PCM that precede the failing packet), ...] in stream order for
        the caller to raise where the reference would have thrown.  Any other failure raises here."""
        self.parse_errors = []
        if rc == native.OK:
            return
        n = C.c_int(0)
        check(lib().nvh_stream_parse_errors(self._h, None, None, 0, C.byref(n)), "nvh_stream_parse_errors")
        if n.value <= 0:
            raise native.NvhError(rc, where)
        codes, before = np.zeros(n.value, np.int32), np.zeros(n.value, np.int64)
        check(lib().nvh_stream_parse_errors(self._h, codes.ctypes.data_as(C.POINTER(C.c_int32)), before.ctypes.data_as(C.POINTER(C.c_int64)),
                                            n.value, C.byref(n)), "nvh_stream_parse_errors")
        self.parse_errors = [(native.NvhError(int(c), where), int(b) * self.channels) for c, b in zip(codes, before)]

    def upload_batch(self):
        h = C.c_void_p()
        check(lib().nvh_batch_upload(self._h, C.byref(h)), "nvh_batch_upload")
        return Batch(h, self)

    def close(self):
        if self._h:
            lib().nvh_stream_close(self._h)
            self._h = C.c_void_p()
        if getattr(self, "_pin_ptr", None):
            self._pin_arr = None
            lib().nvh_pinned_free(self._pin_ptr)
            self._pin_ptr, self._pin_cap = None, 0
        for b in getattr(self, "_pipe", None) or []:
            if b[0]:
                b[2] = None
                lib().nvh_pinned_free(b[0])
                b[0], b[1] = None, 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StreamDecoder:
    """IStreamDecoder-shaped object (Contracts/IStreamDecoder.cs:9-105) over a packet list."""

    def __init__(self, ctx, packets, granules=None, flags=None, batch_frames=1024, gpu_parse=False):
        if len(packets) < 3:
            raise native.NvhError(native.ERR_NOT_VORBIS, "StreamDecoder")
        self._stream = Stream(ctx, packets[0], packets[1], packets[2])
        if gpu_parse:  # packets parsed by k_parse; stream shapes outside its limits silently keep the host parser
            try:
                self._stream.set_gpu_parse(True)
            except native.NvhError as e:
                if e.code != native.ERR_UNSUPPORTED:
                    raise
        self._ctx = ctx
        self._gpu_parse = bool(gpu_parse)
        self._index = None
        self._skip = 0  # floats to discard in front of the next samples (roll-forward after a seek)
        self._packets = packets
        self._array = packets if isinstance(packets, PacketArray) else None  # batched push, no per-packet FFI call
        if self._array is not None and granules is None and flags is None:
            self._granules, self._flags = packets.granules, packets.flags
        else:
            self._granules = granules if granules is not None else [-1] * len(packets)
            self._flags = flags if flags is not None else [0] * len(packets)
        self._next = 3
        self._batch_frames = int(batch_frames)
        self._ring = np.zeros(0, dtype=np.float32)
        self._ring_pos = 0
        self._ended = False
        self._position = 0
        # An exception the reference would throw from inside Read surfaces here at the same place in the PCM: the
        # look-ahead batch is synthesised up to the failing packet's position first.  (error, ring index it belongs at)
        self._pending_errors = []  # [(error, ring index)], ascending

    UpperBitrate = property(lambda self: self._stream.bitrates()[0])
    NominalBitrate = property(lambda self: self._stream.bitrates()[1])
    LowerBitrate = property(lambda self: self._stream.bitrates()[2])
    Channels = property(lambda self: self._stream.channels)
    SampleRate = property(lambda self: self._stream.sample_rate)
    HasClipped = property(lambda self: self._stream.has_clipped())

    @property
    def ClipSamples(self):
        return self._clip if hasattr(self, "_clip") else True

    @ClipSamples.setter
    def ClipSamples(self, on):
        self._clip = bool(on)
        self._stream.set_clip(on)

    @property
    def IsEndOfStream(self):
        return self._ended and self._ring_pos >= self._ring.size

    @property
    def SamplePosition(self):
        pos, _, _ = self._stream.position()
        pending = self._stream.pending()[1]  # pushed, not yet synthesised (only right after a seek)
        return pos - pending - (self._ring.size - self._ring_pos) // self.Channels + self._skip // self.Channels

    def _refill(self):
        """Parse up to batch_frames packets ahead and synthesise them."""
        if self._pending_errors:  # left over from a batch whose ring has been read out
            raise self._pending_errors.pop(0)[0]
        while not self._ended:
            pushed = 0
            push_error = None
            if self._array is not None:
                if self._stream.position()[2]:
                    self._ended = True
                elif self._next >= len(self._array):
                    self._stream.push_end()
                    self._ended = True
                else:
                    try:
                        took = self._stream.push_packets(self._array, self._next, self._batch_frames)
                    except native.NvhError as e:
                        took = getattr(e, "consumed", None)
                        if took is None:
                            raise
                        push_error = e  # surfaces once the frames parsed before it have been read
                    self._next += took
                    if push_error is None and took < self._batch_frames and self._next < len(self._array):
                        self._ended = True  # stopped early: _eosFound
                pushed = self._batch_frames
            while pushed < self._batch_frames:
                if self._stream.position()[2]:
                    self._ended = True  # _eosFound: no more packets are pulled
                    break
                if self._next >= len(self._packets):
                    self._stream.push_end()
                    self._ended = True
                    break
                i = self._next
                self._next += 1
                try:
                    self._stream.push_packet(self._packets[i], self._granules[i], self._flags[i])
                except native.NvhError as e:
                    push_error = e  # the packet is consumed (`_next` already points past it)
                    break
                pushed += 1
            frames, _ = self._stream.pending()
            pcm = None
            if frames:
                # the ring is only replaced once it has been read out, so the stream's pinned buffer can be it
                pcm = self._stream.synth_host(pinned=True)
            got = pcm is not None and pcm.size > 0
            if got:
                self._ring = pcm
                self._ring_pos = 0
            size = pcm.size if got else 0
            if self._stream.parse_errors:  # GPU-parse mode: packets inside the batch failed
                self._pending_errors = [(e, min(at, size)) for e, at in self._stream.parse_errors]
                self._stream.parse_errors = []
            if push_error is not None:  # host-parse mode: everything parsed before the packet comes first
                self._pending_errors.append((push_error, size))
            if got:
                return True
            if self._pending_errors:
                raise self._pending_errors.pop(0)[0]
        return False

    def Read(self, buffer, offset, count):
        """StreamDecoder.Read (StreamDecoder.cs:320-389)."""
        ch = self.Channels
        if offset < 0 or offset + count > len(buffer):
            raise IndexError("offset")  # ArgumentOutOfRangeException
        if count % ch != 0:
            raise ValueError("count must be a multiple of Channels")
        idx, tgt = offset, offset + count
        while idx < tgt:
            if self._pending_errors and self._ring_pos >= self._pending_errors[0][1]:
                # the packet that follows here made the decoder throw; the samples before it have been delivered.  (The
                # reference's Read loses what it copied into the caller's buffer in the same call; so does this.)
                raise self._pending_errors.pop(0)[0]
            if self._ring_pos >= self._ring.size:
                if not self._refill():
                    break
            if self._skip:  # SeekTo's roll-forward into the packet that holds the target (StreamDecoder.cs:625)
                drop = min(self._skip, self._ring.size - self._ring_pos)
                self._ring_pos += drop
                self._skip -= drop
                continue
            take = min(tgt - idx, self._ring.size - self._ring_pos)
            if self._pending_errors:
                take = min(take, self._pending_errors[0][1] - self._ring_pos)
            buffer[idx:idx + take] = self._ring[self._ring_pos:self._ring_pos + take]
            self._ring_pos += take
            idx += take
        return idx - offset

    # ---- seeking (StreamDecoder.cs:562-628) ----
    def _granule_index(self):
        """Granule position after every audio packet, from the geometry-only index of the stream (host work, built once).
        The reference derives the same numbers page by page from the page granule positions and the packets' nominal
        sample counts (Ogg/PacketProvider.cs:74-146); for packets before the first page end they are back-filled from it."""
        if self._index is None:
            pa = PacketArray.from_list(self._packets, self._granules, self._flags)
            pos, em, state, total = self._stream.index_packets(pa, 3)
            gp = pos.copy()
            synced = np.nonzero(state & 4)[0]
            off = int(pos[synced[0]] - em[synced[0]]) if synced.size else 0
            un = (state & 4) == 0
            gp[un] = em[un] + off
            self._index = (gp, state, total + off)
        return self._index

    @property
    def TotalSamples(self):
        """IPacketProvider.GetGranuleCount (StreamDecoder.cs:700): the largest page granule position."""
        g = np.asarray(self._granules[3:] if len(self._granules) > 3 else [], dtype=np.int64)
        g = g[g >= 0]
        if g.size == 0:
            raise native.NvhError(native.ERR_INVALID_DATA, "TotalSamples: no granule positions")
        return int(g.max())

    @property
    def TotalTime(self):
        return self.TotalSamples / float(self.SampleRate)

    def attach_ogg(self, data, stream_index=0):
        """The container the packet list came from: SeekTo then runs the reference's page-level search (nvh_ogg_seek) instead
        of the sample-count index a bare packet list allows."""
        self._ogg = (data, int(stream_index))
        self._ogg_index = None

    def _provider_seek(self, granule_pos, pre_roll):
        """IPacketProvider.SeekTo(granulePos, preRoll, GetPacketGranules) -> (index into the packet list of the packet the
        provider returns next, the granule position the method returns).

        With the Ogg file at hand this is Ogg/PacketProvider.cs:56-72 as the reference runs it once it has read every page:
        page search (Ogg/StreamPageReader.cs:122-264), packet search with the libvorbis granule workaround
        (Ogg/PacketProvider.cs:74-260), NormalizePacketIndex (:262-295) -- including what that does on the first data page,
        where the first packet's nominal length looks like the encoder bug to GetIsVorbisBugDiff (:224-260) and every
        position of the page moves by it.  A bare packet list has no pages: positions then follow the sample counts a serial
        decode produces."""
        if getattr(self, "_ogg", None) is not None:
            L = lib()
            if self._ogg_index is None:
                data, k = self._ogg
                buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
                h = C.c_void_p()
                check(L.nvh_ogg_index_open(buf, len(data), k, C.byref(h)), "nvh_ogg_index_open")
                self._ogg_index = h
            pk, gp = C.c_int64(0), C.c_int64(0)
            rc = L.nvh_ogg_seek(self._ogg_index, self._stream._h, int(granule_pos), int(pre_roll), C.byref(pk), C.byref(gp))
            if rc == native.ERR_ARGUMENT:
                raise IndexError("granulePos")  # ArgumentOutOfRangeException
            check(rc, "nvh_ogg_seek")
            return int(pk.value), int(gp.value)
        gp, state, end_pos = self._granule_index()
        s = int(granule_pos)
        if gp.size == 0:
            raise IndexError("granulePos")
        if s == 0 or s == int(gp[0]):
            return 3, (0 if s == 0 else int(gp[0]))
        if s < int(gp[0]) or s > end_pos:
            raise IndexError("granulePos")
        j = int(np.searchsorted(gp, s, side="left"))  # first packet whose samples reach position s
        if j >= gp.size:
            j = gp.size - 1  # inside the drained tail of the last block
            while j > 0 and gp[j - 1] >= s:
                j -= 1
        return 3 + j - pre_roll, int(gp[j - 1])

    def _push_one(self, i):
        """ReadNextPacket for packet i of the list: True when it decoded (a frame is pending), with its emitted samples."""
        if i >= len(self._packets):
            self._stream.push_end()
            return False, 0
        f0, s0 = self._stream.pending()
        self._stream.push_packet(self._packets[i], self._granules[i], self._flags[i])
        f1, s1 = self._stream.pending()
        decoded = f1 > f0 and int(self._stream.pending_geometry()[-1][0]) != 0  # (a rejected packet drains the previous block: n = 0)
        return decoded, s1 - s0

    def SeekTo(self, sample_position, origin="begin"):
        """StreamDecoder.SeekTo(long, SeekOrigin) (StreamDecoder.cs:562-628): position the provider one packet early
        (:593-598), ResetDecoder, read the pre-roll packet and the packet that holds the target (:603-621), roll forward
        inside it (:624-625).  origin: "begin", "current" (SamplePosition - value, as the reference computes it), "end"
        (TotalSamples - value).  IndexError = ArgumentOutOfRangeException, RuntimeError = InvalidOperationException.

        One place cannot be mirrored: when the roll-forward is longer than the packet's output (possible on the first data
        page, see _provider_seek) the reference's next Read never returns (copyLen < 0 with start != end,
        StreamDecoder.cs:341-377); this raises NvhError(ERR_RUNTIME) instead."""
        if not getattr(self, "can_seek", True):
            raise RuntimeError("Seek is not supported by the Contracts.IPacketProvider instance.")  # InvalidOperationException (:565)
        s = int(sample_position)
        if origin == "current":
            s = self.SamplePosition - s
        elif origin == "end":
            s = self.TotalSamples - s
        elif origin != "begin":
            raise IndexError("origin")
        if s < 0:
            raise IndexError("samplePosition")
        old_pos = self.SamplePosition  # ResetDecoder leaves _currentPosition alone (StreamDecoder.cs:294-305)
        if s == 0:
            k, _ = self._provider_seek(0, 0)  # "short circuit for the looping case" (:587-592)
            roll = 0
            pos = 0
        else:
            k, pos = self._provider_seek(s, 1)
            roll = s - pos
        self._stream.reset()
        self._ring = np.zeros(0, dtype=np.float32)
        self._ring_pos = 0
        self._pending_errors = []
        self._ended = False
        self._skip = 0
        self._stream.set_position_state(True, old_pos)  # _hasPosition = true (:600)
        ok, _ = self._push_one(k)
        self._next = k + 1
        if not ok:
            self._ended = True  # _eosFound: "we'll use this to force ReadSamples to fail to read"
            self._stream.drop_pending()
            if self.TotalSamples != s:
                raise RuntimeError("Could not read pre-roll packet!  Try seeking again prior to reading more samples.")
            self._stream.set_position_state(True, s)
            return
        ok, count = self._push_one(k + 1)
        self._next = k + 2
        if not ok:
            self._stream.reset()
            self._ended = True
            raise RuntimeError("Could not read pre-roll packet!  Try seeking again prior to reading more samples.")
        if roll > count or roll < 0:
            self._stream.reset()
            self._ended = True
            raise native.NvhError(native.ERR_RUNTIME, "SeekTo: roll-forward of %d samples into a packet that emits %d: the "
                                  "reference's Read does not return from here" % (roll, count))
        self._skip = roll * self.Channels
        # _currentPosition = samplePosition (:626): the pending frame's `count` samples are ahead of it
        self._stream.set_position_state(True, s - roll + count)

    def close(self):
        if getattr(self, "_ogg_index", None):
            lib().nvh_ogg_index_close(self._ogg_index)
            self._ogg_index = None
        self._stream.close()


class VorbisReader:
    """VorbisReader-shaped facade (VorbisReader.cs): first logical stream of an .ogg file or byte string."""

    def __init__(self, source, ctx=None, device=0, batch_frames=8192, gpu_parse=True, forward_only=False):
        # gpu_parse: parse the packets on the GPU too when the stream shape allows it (StreamDecoder falls back silently)
        # forward_only: read the container the way the reference reads a source that cannot seek (ContainerReader picks
        # ForwardOnlyPageReader for !stream.CanSeek, Ogg/ContainerReader.cs); SeekTo then raises as IPacketProvider.CanSeek is false
        self._forward_only = bool(forward_only)
        if isinstance(source, (bytes, bytearray, memoryview)):
            data = bytes(source)
        else:
            with open(source, "rb") as fh:
                data = fh.read()
        self._own_ctx = ctx is None
        self._ctx = ctx if ctx is not None else Context(device)
        self._data, self._batch_frames, self._gpu_parse = data, batch_frames, gpu_parse
        # VorbisReader keeps one decoder per logical Vorbis stream of the container and starts on the first
        # (VorbisReader.cs:47-63, 74-87); streams that are not Vorbis are passed over
        self._stream_ids = []
        for k in range(max(1, ogg_stream_count(data))):
            if self._is_vorbis(k):
                self._stream_ids.append(k)
        if not self._stream_ids:
            if self._own_ctx:
                self._ctx.close()
            raise native.NvhError(native.ERR_NOT_VORBIS, "VorbisReader")  # ArgumentException: could not load the container
        self._stream_index = 0
        self._dec = StreamDecoder(self._ctx, demux_ogg_array(data, self._stream_ids[0], self._forward_only), None, None, batch_frames, gpu_parse)
        if self._forward_only:
            self._dec.can_seek = False
        else:
            self._dec.attach_ogg(data, self._stream_ids[0])
        self._decs = {0: self._dec}  # one decoder per logical stream, created on first use, kept like VorbisReader._decoders

    def _is_vorbis(self, k):
        try:
            pa = demux_ogg_array(self._data, k)
            if len(pa) < 3:
                return False
            Stream(None, pa[0], pa[1], pa[2]).close()  # header parse only, on the host
            return True
        except native.NvhError:
            return False

    StreamCount = property(lambda self: len(self._stream_ids))
    StreamIndex = property(lambda self: self._stream_index)

    def SwitchStreams(self, index):
        """VorbisReader.SwitchStreams (VorbisReader.cs:300-322): make logical Vorbis stream `index` the one ReadSamples
        decodes; returns True when its channel count or sample rate differs from the previous stream's."""
        if index < 0 or index >= len(self._stream_ids):
            raise IndexError("index")  # ArgumentOutOfRangeException
        if index == self._stream_index:
            return False
        old = (self.Channels, self.SampleRate)
        clip = self.ClipSamples
        if index not in self._decs:
            self._decs[index] = StreamDecoder(self._ctx, demux_ogg_array(self._data, self._stream_ids[index], self._forward_only), None,
                                              None, self._batch_frames, self._gpu_parse)
            if self._forward_only:
                self._decs[index].can_seek = False
            else:
                self._decs[index].attach_ogg(self._data, self._stream_ids[index])
        self._dec = self._decs[index]
        self._dec.ClipSamples = clip  # carry-through the clipping setting
        self._stream_index = index
        return (self.Channels, self.SampleRate) != old

    Channels = property(lambda self: self._dec.Channels)
    SampleRate = property(lambda self: self._dec.SampleRate)
    IsEndOfStream = property(lambda self: self._dec.IsEndOfStream)
    HasClipped = property(lambda self: self._dec.HasClipped)
    TotalSamples = property(lambda self: self._dec.TotalSamples)
    TotalTime = property(lambda self: self._dec.TotalTime)

    @property
    def SamplePosition(self):
        return self._dec.SamplePosition

    @SamplePosition.setter
    def SamplePosition(self, value):  # VorbisReader.SamplePosition set => SeekTo (StreamDecoder.cs:714-718)
        self._dec.SeekTo(value)

    @property
    def TimePosition(self):
        return self._dec.SamplePosition / float(self.SampleRate)

    @TimePosition.setter
    def TimePosition(self, seconds):  # SeekTo(TimeSpan): (long)(SampleRate * TotalSeconds) (StreamDecoder.cs:552-555)
        self._dec.SeekTo(int(self.SampleRate * float(seconds)))

    def SeekTo(self, sample_position, origin="begin"):
        self._dec.SeekTo(sample_position, origin)

    @property
    def ClipSamples(self):
        return self._dec.ClipSamples

    @ClipSamples.setter
    def ClipSamples(self, on):
        self._dec.ClipSamples = on

    def ReadSamples(self, buffer, offset=0, count=None):
        """VorbisReader.ReadSamples(float[], int, int) (VorbisReader.cs:336-345)."""
        if count is None:
            count = len(buffer) - offset
        count -= count % self.Channels
        if count > 0:
            return self._dec.Read(buffer, offset, count)
        return 0

    def read_all(self):
        chunks = []
        buf = np.empty(65536 * self.Channels, dtype=np.float32)
        while True:
            n = self.ReadSamples(buf, 0, buf.size)
            if n <= 0:
                break
            chunks.append(buf[:n].copy())
        return np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.float32)

    def close(self):
        for d in self._decs.values():
            d.close()
        self._decs = {}
        if self._own_ctx:
            self._ctx.close()
