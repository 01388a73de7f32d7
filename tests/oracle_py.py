"""ctypes wrapper of the CPU oracle (oracle/libnvorbis_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libnvorbis_oracle.so")


def build(force=False):
    srcs = [f for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h", ".inc"))]
    stale = force or not os.path.exists(LIB) or any(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB) for f in srcs)
    if stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"] + (["-B"] if force else []))
    return LIB


class Oracle:
    def __init__(self, lib):
        self.L = L = lib
        vp = C.c_void_p
        L.orc_open_ogg.restype = vp
        L.orc_open_ogg.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
        L.orc_open_packets.restype = vp
        L.orc_open_packets.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int)]
        L.orc_close.argtypes = [vp]
        L.orc_ogg_seek.argtypes = [C.c_char_p, C.c_size_t, vp, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_seek_to.argtypes = [vp, C.c_int64]
        L.orc_total_samples.argtypes = [vp]
        L.orc_total_samples.restype = C.c_int64
        L.orc_read_samples.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
        for f in ("orc_channels", "orc_sample_rate", "orc_block0", "orc_block1", "orc_packet_count", "orc_has_clipped",
                  "orc_is_end_of_stream", "orc_last_error", "orc_trace_count"):
            getattr(L, f).argtypes = [vp]
        L.orc_sample_position.argtypes = [vp]
        L.orc_sample_position.restype = C.c_int64
        L.orc_set_clip_samples.argtypes = [vp, C.c_int]
        L.orc_enable_trace.argtypes = [vp, C.c_int]
        L.orc_trace_data.argtypes = [vp]
        L.orc_trace_data.restype = vp
        L.orc_mdct_reverse.argtypes = [vp, C.c_int]
        L.orc_mdct_tables.argtypes = [C.c_int, vp, vp, vp, vp]
        L.orc_calc_window.argtypes = [C.c_int, C.c_int, C.c_int, vp]
        L.orc_calc_overlap.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_inverse_db.restype = C.c_float
        L.orc_inverse_db.argtypes = [C.c_int]
        L.orc_render_line_multi.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orc_render_point.argtypes = [C.c_int] * 5
        L.orc_inverse_couple.argtypes = [vp, vp, C.c_int]
        L.orc_decode_packet_block.argtypes = [vp, vp, C.c_int, vp] + [C.POINTER(C.c_int)] * 4
        L.orc_residue_decode_at.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_int)]
        L.orc_last_residue_calls.argtypes = [vp, vp, vp, C.c_int, C.POINTER(C.c_int)]
        L.orc_mode_info.argtypes = [vp, C.c_int] + [C.POINTER(C.c_int)] * 3
        L.orc_floor0_apply_coeffs.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp, vp, C.c_int]
        L.orc_floor1_apply_posts.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int]
        L.orc_floor_info.argtypes = [vp, C.c_int] + [C.POINTER(C.c_int)] * 3
        L.orc_last_floor_data.argtypes = [vp, C.c_int, C.POINTER(C.c_int), vp, C.POINTER(C.c_int), C.POINTER(C.c_float), vp, C.c_int]
        L.orc_mapping_info.argtypes = [vp, C.c_int, C.POINTER(C.c_int), vp, vp, C.c_int, vp, C.c_int]
        L.orc_coverage_begin.argtypes = [C.c_int, C.c_int]
        L.orc_coverage_end.argtypes = [vp]

    # ---- transforms / tables ----
    def mdct_reverse(self, x, n):
        buf = np.zeros(n, dtype=np.float32)
        buf[: n // 2] = x[: n // 2]
        self.L.orc_mdct_reverse(buf.ctypes.data, n)
        return buf

    def mdct_tables(self, n):
        a, b = np.zeros(n // 2, np.float32), np.zeros(n // 2, np.float32)
        c, br = np.zeros(n // 4, np.float32), np.zeros(n // 8, np.uint16)
        self.L.orc_mdct_tables(n, a.ctypes.data, b.ctypes.data, c.ctypes.data, br.ctypes.data)
        return a, b, c, br

    def window(self, prev, block, nxt):
        w = np.zeros(block, np.float32)
        self.L.orc_calc_window(prev, block, nxt, w.ctypes.data)
        return w

    def overlap(self, prev, block, nxt):
        s, v, t = C.c_int(0), C.c_int(0), C.c_int(0)
        self.L.orc_calc_overlap(prev, block, nxt, C.byref(s), C.byref(v), C.byref(t))
        return s.value, v.value, t.value

    # ---- single packets ----
    def open_headers(self, headers):
        """Decoder handle over the three header packets only (for orc_decode_packet_block); close with L.orc_close."""
        blob = np.frombuffer(b"".join(headers), dtype=np.uint8)
        offs = np.zeros(4, np.int64)
        offs[1:] = np.cumsum([len(p) for p in headers])
        g3, f3, err = np.full(3, -1, np.int64), np.zeros(3, np.uint8), C.c_int(0)
        d = self.L.orc_open_packets(blob.ctypes.data, offs.ctypes.data, g3.ctypes.data, f3.ctypes.data, 3, C.byref(err))
        if not d:
            raise RuntimeError("oracle open failed: %d" % err.value)
        return d

    def packet_coverage(self, d, pkt):
        """Decode one packet (Mode.Decode) with the residue coverage hook armed.
        Returns (block_size, mask[channels][block1]) -- mask bit s set where cascade stage s added a value -- or None."""
        ch, b1 = self.L.orc_channels(d), self.L.orc_block1(d)
        planes = np.zeros(ch * b1, np.float32)
        mask = np.zeros((ch, b1), np.uint8)
        a, b, c, e = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert self.L.orc_coverage_begin(ch, b1) == 0
        rc = self.L.orc_decode_packet_block(d, pkt, len(pkt), planes.ctypes.data, C.byref(a), C.byref(b), C.byref(c), C.byref(e))
        self.L.orc_coverage_end(mask.ctypes.data)
        if rc != 1:
            return None
        return e.value, mask

    # ---- seeking (Ogg/PacketProvider.cs:56-295, StreamDecoder.cs:562-628) ----
    def open_ogg(self, data):
        err = C.c_int(0)
        d = self.L.orc_open_ogg(data, len(data), C.byref(err))
        if not d:
            raise RuntimeError("oracle open failed: %d" % err.value)
        return d

    def ogg_seek(self, data, d, granule_pos, pre_roll):
        """IPacketProvider.SeekTo -> (rc, packet index in the demuxed list, returned granule position)."""
        k, g = C.c_int64(0), C.c_int64(0)
        rc = self.L.orc_ogg_seek(data, len(data), d, int(granule_pos), int(pre_roll), C.byref(k), C.byref(g))
        return rc, int(k.value), int(g.value)

    def seek_and_read(self, d, sample_position, nfloats, clip=True):
        """StreamDecoder.SeekTo then one ReadSamples: (rc of the seek, samples or the read's negative code, position after)."""
        L = self.L
        rc = L.orc_seek_to(d, int(sample_position))
        if rc != 0:
            return rc, None, int(L.orc_sample_position(d))
        L.orc_set_clip_samples(d, 1 if clip else 0)
        buf = np.zeros(nfloats, np.float32)
        n = L.orc_read_samples(d, buf.ctypes.data, buf.size, 0, buf.size)
        return 0, (buf[:n].copy() if n >= 0 else n), int(L.orc_sample_position(d))

    # ---- decoding ----
    def decode_ogg(self, data, clip=True, chunk=4096, trace=False):
        err = C.c_int(0)
        d = self.L.orc_open_ogg(data, len(data), C.byref(err))
        if not d:
            raise RuntimeError("oracle open failed: %d" % err.value)
        return self._drain(d, clip, chunk, trace)

    def decode_packets(self, packets, granules, flags, clip=True, chunk=4096, trace=False):
        blob = b"".join(packets)
        offs = np.zeros(len(packets) + 1, np.int64)
        offs[1:] = np.cumsum([len(p) for p in packets])
        gr = np.asarray(granules, np.int64)
        fl = np.asarray(flags, np.uint8)
        bb = np.frombuffer(blob if blob else b"\0", dtype=np.uint8)
        err = C.c_int(0)
        d = self.L.orc_open_packets(bb.ctypes.data, offs.ctypes.data, gr.ctypes.data, fl.ctypes.data, len(packets), C.byref(err))
        if not d:
            raise RuntimeError("oracle open failed: %d" % err.value)
        return self._drain(d, clip, chunk, trace)

    def _drain(self, d, clip, chunk, trace):
        L = self.L
        try:
            ch = L.orc_channels(d)
            L.orc_set_clip_samples(d, 1 if clip else 0)
            if trace:
                L.orc_enable_trace(d, 1)
            buf = np.zeros(chunk * ch, np.float32)
            out = []
            while True:
                n = L.orc_read_samples(d, buf.ctypes.data, buf.size, 0, buf.size)
                if n < 0:
                    raise RuntimeError("oracle read failed: %d" % n)
                if n == 0:
                    break
                out.append(buf[:n].copy())
            pcm = np.concatenate(out) if out else np.zeros(0, np.float32)
            info = dict(channels=ch, sample_rate=L.orc_sample_rate(d), block0=L.orc_block0(d), block1=L.orc_block1(d),
                        has_clipped=bool(L.orc_has_clipped(d)), position=L.orc_sample_position(d))
            if trace:
                tn = L.orc_trace_count(d)
                info["trace"] = np.ctypeslib.as_array(C.cast(L.orc_trace_data(d), C.POINTER(C.c_int32)), shape=(tn, 6)).copy()
            return pcm, info
        finally:
            L.orc_close(d)


_cached = None


def load():
    global _cached
    if _cached is None:
        _cached = Oracle(C.CDLL(build()))
    return _cached
