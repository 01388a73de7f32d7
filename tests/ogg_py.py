"""Minimal Ogg page writer / reader for tests (RFC 3533 framing; test infrastructure).

Writer: laces packets into CRC-valid pages (255-byte segments, continued packets across pages, BOS/EOS flags,
granule position = position after the last packet that *finishes* on the page, -1 if none does).
Reader: plain page walk without error recovery (used to cross-check the product's and the oracle's demux).
"""
import struct
import zlib

_REV8 = bytes(int("{:08b}".format(i)[::-1], 2) for i in range(256))


def _rev32(v):
    return int("{:032b}".format(v)[::-1], 2)


def ogg_crc(data):
    """CRC-32 with polynomial 0x04c11db7, initial value 0, no reflection, no final xor (RFC 3533 section 6).

    Computed with zlib's reflected CRC-32 engine on bit-reversed bytes: reflecting input bytes and the register turns
    one convention into the other; starting zlib at 0xFFFFFFFF cancels its initial and final inversions."""
    return _rev32((zlib.crc32(bytes(data).translate(_REV8), 0xFFFFFFFF) ^ 0xFFFFFFFF) & 0xFFFFFFFF)


def _crc_slow(data):
    crc = 0
    for b in data:
        crc ^= b << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if crc & 0x80000000 else (crc << 1) & 0xFFFFFFFF
    return crc


def make_page(serial, seq, granule, flags, segments, body):
    hdr = struct.pack("<4sBBqIII", b"OggS", 0, flags, granule, serial, seq, 0) + bytes([len(segments)]) + bytes(segments)
    crc = ogg_crc(hdr + body)
    return hdr[:22] + struct.pack("<I", crc) + hdr[26:] + body


class PageWriter:
    """Laces packets of one logical stream into pages."""

    def __init__(self, serial, max_segments=255):
        self.serial = serial
        self.seq = 0
        self.max_segments = max_segments
        self.pages = []
        self._segs = []
        self._body = bytearray()
        self._granule = -1
        self._continued = False
        self._bos = True

    def _flush(self, eos=False, next_continued=False):
        if not self._segs and not eos:
            return
        flags = (1 if self._continued else 0) | (2 if self._bos else 0) | (4 if eos else 0)
        self.pages.append(make_page(self.serial, self.seq, self._granule, flags, self._segs, bytes(self._body)))
        self.seq += 1
        self._bos = False
        self._segs = []
        self._body = bytearray()
        self._granule = -1
        self._continued = next_continued

    def add_packet(self, data, granule, flush=False, eos=False):
        n = len(data)
        lacing = [255] * (n // 255) + [n % 255]
        if len(self._segs) + len(lacing) <= self.max_segments:  # the whole packet fits the open page: no flush inside it
            self._segs += lacing
            self._body += data
            self._granule = granule
            if flush or eos:
                self._flush(eos=eos)
            return
        pos = 0
        for k, seg in enumerate(lacing):
            if len(self._segs) == self.max_segments:
                self._flush(next_continued=k > 0)
            self._segs.append(seg)
            self._body += data[pos:pos + seg]
            pos += seg
        self._granule = granule
        if flush or eos:
            self._flush(eos=eos)

    def finish(self):
        self._flush()
        return b"".join(self.pages)


def write_ogg(packets, granules, serial=0x4E56, page_packets=None, max_segments=255, eos=True):
    """packets[0:3] = Vorbis headers.  granules[i] = absolute sample position after packet i (ignored for headers).

    Layout as libvorbis writes it: identification header alone on the first (BOS) page, comment + setup on their own
    page(s), audio packets from a fresh page; a page is flushed when it is full or holds `page_packets` packets."""
    pw = PageWriter(serial, max_segments)
    pw.add_packet(packets[0], 0, flush=True)
    pw.add_packet(packets[1], 0)
    pw.add_packet(packets[2], 0, flush=True)
    count = 0
    last = len(packets) - 1
    for i in range(3, len(packets)):
        count += 1
        is_last = i == last
        flush = page_packets is not None and count >= page_packets
        pw.add_packet(packets[i], granules[i], flush=flush and not is_last, eos=is_last and eos)
        if flush:
            count = 0
    return pw.finish()


def read_pages(data):
    """[(flags, granule, serial, seq, crc_ok, [packet segments...], body)] -- no resync, stops at the first non-page."""
    pages = []
    pos = 0
    while pos + 27 <= len(data) and data[pos:pos + 4] == b"OggS":
        _, ver, flags, granule, serial, seq, crc = struct.unpack_from("<4sBBqIII", data, pos)
        nseg = data[pos + 26]
        segs = list(data[pos + 27:pos + 27 + nseg])
        blen = sum(segs)
        end = pos + 27 + nseg + blen
        page = bytearray(data[pos:end])
        page[22:26] = b"\0\0\0\0"
        pages.append(dict(flags=flags, granule=granule, serial=serial, seq=seq, crc_ok=ogg_crc(page) == crc, segs=segs,
                          body=bytes(data[pos + 27 + nseg:end]), offset=pos, length=end - pos))
        pos = end
    return pages


def read_packets(data, serial=None):
    """Packets of one logical stream (the first one found unless `serial` is given): (packets, granules, eos_flags).

    granule of a packet = granule of the page it finishes on if it is the last packet finishing there, else -1."""
    pages = read_pages(data)
    if serial is None and pages:
        serial = pages[0]["serial"]
    packets, granules, eos = [], [], []
    cur = bytearray()
    for pg in pages:
        if pg["serial"] != serial:
            continue
        pos = 0
        done_here = []
        for seg in pg["segs"]:
            cur += pg["body"][pos:pos + seg]
            pos += seg
            if seg < 255:
                packets.append(bytes(cur))
                granules.append(-1)
                eos.append(False)
                done_here.append(len(packets) - 1)
                cur = bytearray()
        if done_here:
            granules[done_here[-1]] = pg["granule"]
            if pg["flags"] & 4:
                eos[done_here[-1]] = True
    return packets, granules, eos
