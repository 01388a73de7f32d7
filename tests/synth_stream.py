"""Synthetic Vorbis stream generator for parity tests (test infrastructure).

The shipped TestFiles only exercise Floor1 + Residue1/2 with 1-2 channels and 256/2048 blocks.  This module
writes *setup headers* for arbitrary configurations (Floor0, Residue0, >2 channels, several coupling steps,
several submaps, block sizes 64..8192, codebooks whose dimension does not divide the partition size, ...)
and fills the audio packets with random bits.  Every codebook is a complete fixed-length prefix code (entries
a power of two), so any bit string decodes to *some* side information: no Huffman encoder is needed, and the
oracle and the product are compared on whatever the bits mean.  Header layouts follow what the reference
parses (StreamDecoder.cs:179-289, Codebook.cs:59-283, Floor0.cs:28-65, Floor1.cs:30-133, Residue0.cs:35-117,
Mapping.cs:16-93, Mode.cs:24-67).
"""
import math

import numpy as np


class BitWriter:
    """LSB-first bit packer (the inverse of DataPacket.ReadBits, DataPacket.cs:150-283)."""

    def __init__(self):
        self.acc = 0
        self.nbits = 0
        self.out = bytearray()

    def write(self, value, bits):
        if bits == 0:
            return
        assert 0 <= value < (1 << bits), (value, bits)
        self.acc |= value << self.nbits
        self.nbits += bits
        while self.nbits >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.nbits -= 8

    def bytes(self):
        out = bytearray(self.out)
        if self.nbits:
            out.append(self.acc & 0xFF)
        return bytes(out)


def ilog(x):
    n = 0
    while x > 0:
        n += 1
        x >>= 1
    return n


def vorbis_float(mantissa, exponent):
    """Pack mantissa * 2**exponent the way Utils.ConvertFromVorbisFloat32 unpacks it (Utils.cs:45-59)."""
    sign = 0
    if mantissa < 0:
        sign = 1
        mantissa = -mantissa
    assert mantissa < (1 << 21)
    e = exponent + 788
    assert 0 <= e < 1024
    return (sign << 31) | (e << 21) | mantissa


class Book:
    """Fixed-length complete codebook: `bits` code bits, 2**bits entries."""

    def __init__(self, bits, dims=1, lookup=0, min_me=(0, 0), delta_me=(1, 0), value_bits=4, sequence_p=0, mults=None):
        self.bits, self.dims, self.lookup = bits, dims, lookup
        self.entries = 1 << bits
        self.min_me, self.delta_me, self.value_bits, self.sequence_p, self.mults = min_me, delta_me, value_bits, sequence_p, mults

    def lookup1_values(self):
        r = int(math.floor(math.exp(math.log(self.entries) / self.dims)))
        if math.floor((r + 1) ** self.dims) <= self.entries:
            r += 1
        return r

    def write(self, w, rng):
        w.write(0x564342, 24)
        w.write(self.dims, 16)
        w.write(self.entries, 24)
        w.write(0, 1)  # not ordered
        w.write(0, 1)  # not sparse
        for _ in range(self.entries):
            w.write(self.bits - 1, 5)
        w.write(self.lookup, 4)
        if self.lookup == 0:
            return
        w.write(vorbis_float(*self.min_me), 32)
        w.write(vorbis_float(*self.delta_me), 32)
        w.write(self.value_bits - 1, 4)
        w.write(self.sequence_p, 1)
        count = self.lookup1_values() if self.lookup == 1 else self.entries * self.dims
        mults = self.mults if self.mults is not None else rng.integers(0, 1 << self.value_bits, count).tolist()
        assert len(mults) == count, (len(mults), count)
        for m in mults:
            w.write(int(m), self.value_bits)


class IncompleteBook(Book):
    """Like Book, but written sparse with its last entry unused: the all-ones code is unassigned, and a packet that
    contains it makes the reference fault (no overflow list to consult, Codebook.cs:306 -> NullReferenceException)."""

    def write(self, w, rng):
        w.write(0x564342, 24)
        w.write(self.dims, 16)
        w.write(self.entries, 24)
        w.write(0, 1)  # not ordered
        w.write(1, 1)  # sparse
        for i in range(self.entries):
            used = i != self.entries - 1
            w.write(1 if used else 0, 1)
            if used:
                w.write(self.bits - 1, 5)
        w.write(self.lookup, 4)
        if self.lookup == 0:
            return
        w.write(vorbis_float(*self.min_me), 32)
        w.write(vorbis_float(*self.delta_me), 32)
        w.write(self.value_bits - 1, 4)
        w.write(self.sequence_p, 1)
        count = self.lookup1_values() if self.lookup == 1 else self.entries * self.dims
        mults = self.mults if self.mults is not None else rng.integers(0, 1 << self.value_bits, count).tolist()
        for m in mults:
            w.write(int(m), self.value_bits)


def write_floor1(w, partition_classes, class_dims, class_subclass_bits, masterbooks, subclass_books, multiplier, rangebits, xs):
    """Floor1.Init layout (Floor1.cs:30-92).  xs: the X values after the two implicit ones."""
    w.write(1, 16)
    w.write(len(partition_classes), 5)
    for c in partition_classes:
        w.write(c, 4)
    for c in range(max(partition_classes) + 1):
        w.write(class_dims[c] - 1, 3)
        w.write(class_subclass_bits[c], 2)
        if class_subclass_bits[c] > 0:
            w.write(masterbooks[c], 8)
        for j in range(1 << class_subclass_bits[c]):
            w.write(subclass_books[c][j] + 1, 8)  # -1 => unused
    w.write(multiplier - 1, 2)
    w.write(rangebits, 4)
    for x in xs:
        w.write(x, rangebits)


def write_floor0(w, order, rate, bark_map_size, amp_bits, amp_ofs, books):
    """Floor0.Init layout (Floor0.cs:28-49)."""
    w.write(0, 16)
    w.write(order, 8)
    w.write(rate, 16)
    w.write(bark_map_size, 16)
    w.write(amp_bits, 6)
    w.write(amp_ofs, 8)
    w.write(len(books) - 1, 4)
    for b in books:
        w.write(b, 8)


def write_residue(w, rtype, begin, end, psize, classbook, cascades, books):
    """Residue0.Init layout (Residue0.cs:35-66).  books: flat list in (class, stage) order of set cascade bits."""
    w.write(rtype, 16)
    w.write(begin, 24)
    w.write(end, 24)
    w.write(psize - 1, 24)
    w.write(len(cascades) - 1, 6)
    w.write(classbook, 8)
    for c in cascades:
        w.write(c & 7, 3)
        if c >> 3:
            w.write(1, 1)
            w.write(c >> 3, 5)
        else:
            w.write(0, 1)
    for b in books:
        w.write(b, 8)


def write_mapping(w, channels, submaps, coupling, mux, submap_floor_residue):
    """Mapping.Init layout (Mapping.cs:16-78)."""
    w.write(0, 16)
    if submaps > 1:
        w.write(1, 1)
        w.write(submaps - 1, 4)
    else:
        w.write(0, 1)
    if coupling:
        w.write(1, 1)
        w.write(len(coupling) - 1, 8)
        bits = ilog(channels - 1)
        for mag, ang in coupling:
            w.write(mag, bits)
            w.write(ang, bits)
    else:
        w.write(0, 1)
    w.write(0, 2)
    if submaps > 1:
        for c in range(channels):
            w.write(mux[c], 4)
    for fl, rs in submap_floor_residue:
        w.write(0, 8)
        w.write(fl, 8)
        w.write(rs, 8)


def id_header(channels, rate, block0, block1):
    w = BitWriter()
    for b in b"\x01vorbis":
        w.write(b, 8)
    w.write(0, 32)
    w.write(channels, 8)
    w.write(rate, 32)
    w.write(0, 32)
    w.write(0, 32)
    w.write(0, 32)
    w.write(ilog(block0) - 1, 4)
    w.write(ilog(block1) - 1, 4)
    w.write(1, 1)
    return w.bytes()


def comment_header():
    w = BitWriter()
    for b in b"\x03vorbis":
        w.write(b, 8)
    w.write(0, 32)
    w.write(0, 32)
    w.write(1, 1)
    return w.bytes()


def make_stream(cfg, npackets, seed, consistent_windows=True, min_len=16, max_len=220):
    """cfg: dict, see CONFIGS.  Returns (packets, granules, flags): 3 headers + npackets random audio packets."""
    rng = np.random.default_rng(seed)
    ch, b0, b1 = cfg["channels"], cfg["block0"], cfg["block1"]
    w = BitWriter()
    for b in b"\x05vorbis":
        w.write(b, 8)
    books = cfg["books"]
    w.write(len(books) - 1, 8)
    for bk in books:
        bk.write(w, rng)
    w.write(0, 6)   # time count - 1
    w.write(0, 16)
    w.write(len(cfg["floors"]) - 1, 6)
    for f in cfg["floors"]:
        f(w)
    w.write(len(cfg["residues"]) - 1, 6)
    for r in cfg["residues"]:
        r(w)
    w.write(len(cfg["mappings"]) - 1, 6)
    for m in cfg["mappings"]:
        m(w)
    modes = cfg["modes"]  # list of (blockflag, mapping)
    w.write(len(modes) - 1, 6)
    for flag, mp in modes:
        w.write(flag, 1)
        w.write(0, 16)
        w.write(0, 16)
        w.write(mp, 8)
    w.write(1, 1)
    packets = [id_header(ch, cfg.get("rate", 44100), b0, b1), comment_header(), w.bytes()]
    mode_bits = ilog(len(modes) - 1)
    prev_long = None
    seq = rng.integers(0, len(modes), npackets)
    for k in range(npackets):
        pw = BitWriter()
        pw.write(0, 1)
        m = int(seq[k])
        pw.write(m, mode_bits)
        is_long = modes[m][0] == 1
        if is_long:
            if consistent_windows:
                nxt_long = modes[int(seq[k + 1])][0] == 1 if k + 1 < npackets else True
                pf = 1 if (prev_long is None or prev_long) else 0
                pw.write(pf, 1)
                pw.write(1 if nxt_long else 0, 1)
            else:
                pw.write(int(rng.integers(0, 2)), 1)
                pw.write(int(rng.integers(0, 2)), 1)
        prev_long = is_long
        body = rng.integers(0, 256, int(rng.integers(min_len, max_len))).astype(np.uint8)
        # bias: floor "has energy" bits etc. are just random; sprinkle zeros so that some channels stay silent
        if rng.random() < 0.15:
            body[: int(rng.integers(1, 4))] = 0
        for byte in body.tolist():
            pw.write(byte, 8)
        packets.append(pw.bytes())
    return packets, [-1] * len(packets), [0] * len(packets)


# ---- configurations --------------------------------------------------------------------------------

def _floor1_small(class_book, sub_book):
    # 2 partitions of class 0 (dim 2, 1 subclass bit): 6 posts in total
    return lambda w: write_floor1(w, [0, 0], {0: 2}, {0: 1}, {0: class_book}, {0: [sub_book, sub_book]}, 1, 7, [32, 96, 16, 64])


def _floor1_long(class_book, sub_book, rangebits, n_parts=5):
    xs = []
    step = (1 << rangebits) // (2 * n_parts * 2 + 1)
    for i in range(2 * n_parts):
        xs.append(step * (i + 1) + (i % 3))
    return lambda w: write_floor1(w, [0] * n_parts, {0: 2}, {0: 1}, {0: class_book}, {0: [sub_book, -1]}, 2, rangebits, xs)


def config(name):
    # common books: 0 = 1-bit scalar (masterbook / class word helper), 1 = 2-bit scalar (floor posts 0..3),
    # 2 = 4-bit classbook (dim 2, 4 classes), 3 = VQ dim2 16 entries lattice, 4 = VQ dim4 256 entries lattice,
    # 5 = VQ dim8 256 entries lattice, 6 = VQ dim3 8 entries explicit (type 2), 7 = VQ dim2 sequence_p lattice,
    # 8 = VQ dim1 4 entries, 9 = VQ dim5 32 entries type 2
    books = [
        Book(1), Book(2), Book(4, dims=2),
        Book(4, dims=2, lookup=1, min_me=(-3, -2), delta_me=(1, -1), value_bits=3, mults=[0, 2, 4, 6]),
        Book(8, dims=4, lookup=1, min_me=(-5, -4), delta_me=(3, -4), value_bits=3, mults=[0, 1, 2, 3]),
        Book(8, dims=8, lookup=1, min_me=(-1, -3), delta_me=(1, -2), value_bits=2, mults=[0, 1]),
        Book(3, dims=3, lookup=2, min_me=(-7, -5), delta_me=(1, -4), value_bits=4),
        Book(4, dims=2, lookup=1, min_me=(-3, -3), delta_me=(1, -3), value_bits=3, sequence_p=1, mults=[1, 0, 3, 2]),
        Book(2, dims=1, lookup=1, min_me=(-3, -4), delta_me=(1, -3), value_bits=3, mults=[0, 2, 4, 6]),
        Book(5, dims=5, lookup=2, min_me=(-9, -6), delta_me=(1, -5), value_bits=5),
        # floor0 LSP book: dim 4, values in (0, pi): min 0.2, delta ~0.18
        Book(8, dims=4, lookup=1, min_me=(13, -6), delta_me=(45, -6), value_bits=2, mults=[0, 1, 2, 3]),
    ]
    c = dict(books=books, rate=44100)
    if name == "mono_res0_small_blocks":      # generic kernels (n < 256), Residue0, mono
        c.update(channels=1, block0=64, block1=128,
                 floors=[_floor1_small(0, 1)],
                 residues=[lambda w: write_residue(w, 0, 0, 64, 8, 2, [1, 3, 5, 0], [3, 3, 8, 3, 5])],
                 mappings=[lambda w: write_mapping(w, 1, 1, [], None, [(0, 0)])],
                 modes=[(0, 0), (1, 0)])
    elif name == "stereo_res1_coupled":       # Residue1 with two channels and coupling
        c.update(channels=2, block0=256, block1=2048,
                 floors=[_floor1_small(0, 1), _floor1_long(0, 1, 10)],
                 residues=[lambda w: write_residue(w, 1, 0, 128, 16, 2, [1, 2, 7, 0], [3, 4, 3, 4, 5]),
                           lambda w: write_residue(w, 1, 8, 1000, 32, 2, [3, 1, 4, 6], [3, 4, 5, 5, 4, 3])],
                 mappings=[lambda w: write_mapping(w, 2, 1, [(0, 1)], None, [(0, 0)]),
                           lambda w: write_mapping(w, 2, 1, [(1, 0)], None, [(1, 1)])],
                 modes=[(0, 0), (1, 1)])
    elif name == "three_ch_res2_misaligned":  # quirk B-1: partition size not a channel multiple, two coupling steps
        c.update(channels=3, block0=256, block1=1024,
                 floors=[_floor1_small(0, 1), _floor1_long(0, 1, 9)],
                 residues=[lambda w: write_residue(w, 2, 0, 360, 16, 2, [1, 3, 0, 2], [3, 4, 3, 3]),
                           lambda w: write_residue(w, 2, 4, 1500, 32, 2, [1, 3, 5, 2], [4, 3, 5, 3, 4, 5])],
                 mappings=[lambda w: write_mapping(w, 3, 1, [(0, 1), (2, 1)], None, [(0, 0)]),
                           lambda w: write_mapping(w, 3, 1, [(0, 2), (1, 2)], None, [(1, 1)])],
                 modes=[(0, 0), (1, 1)])
    elif name == "six_ch_res2_4096":          # BASELINE config C4 shape: 6 channels, n = 4096, psize 48
        c.update(channels=6, block0=512, block1=4096, rate=48000,
                 floors=[_floor1_small(0, 1), _floor1_long(0, 1, 11, n_parts=8)],
                 residues=[lambda w: write_residue(w, 2, 0, 6 * 200, 48, 2, [1, 3, 0, 2], [3, 4, 3, 3]),
                           lambda w: write_residue(w, 2, 0, 6 * 1536, 48, 2, [1, 3, 5, 7], [4, 3, 5, 3, 4, 5, 3, 4])],
                 mappings=[lambda w: write_mapping(w, 6, 1, [(0, 2), (3, 4)], None, [(0, 0)]),
                           lambda w: write_mapping(w, 6, 1, [(0, 2), (3, 4)], None, [(1, 1)])],
                 modes=[(0, 0), (1, 1)])
    elif name == "floor0_stereo":             # Floor0 (no shipped file uses it)
        c.update(channels=2, block0=256, block1=1024,
                 floors=[lambda w: write_floor0(w, 8, 22050, 64, 5, 40, [10]),
                         lambda w: write_floor0(w, 7, 22050, 128, 6, 30, [10])],
                 residues=[lambda w: write_residue(w, 1, 0, 120, 8, 2, [1, 1, 3, 0], [8, 3, 3, 4])],
                 mappings=[lambda w: write_mapping(w, 2, 1, [(0, 1)], None, [(0, 0)]),
                           lambda w: write_mapping(w, 2, 1, [], None, [(1, 0)])],
                 modes=[(0, 0), (1, 1)])
    elif name == "floor0_slab":               # Floor0 (even and odd order) and a Floor1 mode, residues the slab kernels take
        c.update(channels=2, block0=256, block1=2048,
                 floors=[lambda w: write_floor0(w, 8, 22050, 64, 5, 40, [10]),
                         lambda w: write_floor0(w, 7, 22050, 128, 6, 30, [10]),
                         _floor1_long(0, 1, 10)],
                 residues=[lambda w: write_residue(w, 2, 0, 200, 16, 2, [1, 2, 7, 0], [3, 4, 3, 4, 5]),
                           lambda w: write_residue(w, 2, 0, 1800, 32, 2, [3, 1, 4, 6], [3, 4, 5, 5, 4, 3])],
                 mappings=[lambda w: write_mapping(w, 2, 1, [(0, 1)], None, [(0, 0)]),
                           lambda w: write_mapping(w, 2, 1, [(1, 0)], None, [(1, 1)]),
                           lambda w: write_mapping(w, 2, 1, [(0, 1)], None, [(2, 1)])],
                 modes=[(0, 0), (1, 1), (1, 2)])
    elif name == "two_submaps":               # quirk B-3: every channel ends up ForceNoEnergy
        c.update(channels=2, block0=256, block1=512,
                 floors=[_floor1_small(0, 1), _floor1_small(0, 1)],
                 residues=[lambda w: write_residue(w, 1, 0, 100, 4, 2, [1, 1, 1, 0], [3, 8, 4]),
                           lambda w: write_residue(w, 0, 0, 100, 4, 2, [1, 1, 1, 0], [3, 3, 8])],
                 mappings=[lambda w: write_mapping(w, 2, 2, [(0, 1)], [0, 1], [(0, 0), (1, 1)])],
                 modes=[(0, 0), (1, 0)])
    elif name == "equal_blocks_overrun":      # block0 == block1; book dimension 3 / 5 does not divide the partition size
        c.update(channels=2, block0=1024, block1=1024,
                 floors=[_floor1_long(0, 1, 9)],
                 residues=[lambda w: write_residue(w, 1, 0, 400, 16, 2, [1, 2, 3, 0], [6, 9, 6, 7]),
                           lambda w: write_residue(w, 2, 0, 800, 16, 2, [1, 2, 3, 0], [6, 9, 7, 3])],
                 mappings=[lambda w: write_mapping(w, 2, 1, [(0, 1)], None, [(0, 0)]),
                           lambda w: write_mapping(w, 2, 1, [], None, [(0, 1)])],
                 modes=[(0, 0), (1, 1)])
    elif name == "mono_8192":                 # largest block size
        c.update(channels=1, block0=2048, block1=8192,
                 floors=[_floor1_long(0, 1, 11, n_parts=4), _floor1_long(0, 1, 13, n_parts=10)],
                 residues=[lambda w: write_residue(w, 1, 0, 900, 32, 2, [1, 3, 0, 2], [3, 4, 3, 3]),
                           lambda w: write_residue(w, 2, 0, 4000, 64, 2, [1, 3, 5, 7], [4, 3, 5, 3, 4, 5, 3, 4])],
                 mappings=[lambda w: write_mapping(w, 1, 1, [], None, [(0, 0)]),
                           lambda w: write_mapping(w, 1, 1, [], None, [(1, 1)])],
                 modes=[(0, 0), (1, 1)])
    elif name == "stereo_8192":               # largest block size, two coupled channels: the slab kernels' non-in-place transform
        c.update(channels=2, block0=512, block1=8192,
                 floors=[_floor1_long(0, 1, 9, n_parts=3), _floor1_long(0, 1, 13, n_parts=10)],
                 residues=[lambda w: write_residue(w, 1, 0, 240, 16, 2, [1, 3, 0, 2], [3, 4, 3, 3]),
                           lambda w: write_residue(w, 2, 0, 7936, 64, 2, [1, 3, 5, 7], [4, 3, 5, 3, 4, 5, 3, 4])],
                 mappings=[lambda w: write_mapping(w, 2, 1, [(1, 0)], None, [(0, 0)]),
                           lambda w: write_mapping(w, 2, 1, [(0, 1)], None, [(1, 1)])],
                 modes=[(0, 0), (1, 1)])
    elif name == "mono_res1_2048":            # mono through the slab kernels (blocks 256 / 2048): paired emission with one channel
        c.update(channels=1, block0=256, block1=2048,
                 floors=[_floor1_small(0, 1), _floor1_long(0, 1, 10)],
                 residues=[lambda w: write_residue(w, 1, 0, 128, 16, 2, [1, 2, 7, 0], [3, 4, 3, 4, 5]),
                           lambda w: write_residue(w, 1, 8, 1000, 32, 2, [3, 1, 4, 6], [3, 4, 5, 5, 4, 3])],
                 mappings=[lambda w: write_mapping(w, 1, 1, [], None, [(0, 0)]),
                           lambda w: write_mapping(w, 1, 1, [], None, [(1, 1)])],
                 modes=[(0, 0), (1, 1)])
    elif name in ("res0_slab", "odd_dims_slab", "res2_alias_stereo", "two_pass_slab", "res0_3ch"):
        # the slab kernels' general bin walk (kernels_synth.hip: residue_walk_general): Residue0; lattice books of dimension
        # 1 / 3 / 5; Residue2 over two channels whose partitions (15 components) share bins (quirk B-1); two residue passes
        # per frame (two submaps with the same floor and residue, so that quirk B-3 leaves the channels their energy)
        c["books"] = books + [
            Book(6, dims=3, lookup=1, min_me=(-4, -3), delta_me=(1, -2), value_bits=3, mults=[0, 1, 3, 5]),   # 11
            Book(5, dims=5, lookup=1, min_me=(-2, -3), delta_me=(3, -3), value_bits=2, mults=[0, 3]),         # 12
        ]
        if name in ("res0_slab", "res0_3ch"):  # (res0_3ch: the same through the wide kernel, k_synth8_g)
            res = [lambda w: write_residue(w, 0, 0, 120, 24, 2, [1, 2, 7, 0], [3, 11, 4, 5, 8]),
                   lambda w: write_residue(w, 0, 8, 992, 24, 2, [3, 1, 4, 6], [3, 4, 5, 5, 11, 8])]
        elif name == "odd_dims_slab":
            res = [lambda w: write_residue(w, 1, 0, 120, 15, 2, [1, 2, 7, 0], [8, 11, 12, 11, 8]),
                   lambda w: write_residue(w, 1, 5, 995, 30, 2, [3, 1, 4, 6], [11, 12, 8, 3, 12, 11])]
        elif name == "res2_alias_stereo":
            res = [lambda w: write_residue(w, 2, 3, 243, 15, 2, [1, 2, 7, 0], [8, 11, 12, 11, 8]),
                   lambda w: write_residue(w, 2, 9, 1989, 30, 2, [3, 1, 4, 6], [11, 12, 8, 3, 12, 11])]
        else:
            res = [lambda w: write_residue(w, 1, 0, 128, 16, 2, [1, 2, 7, 0], [3, 4, 3, 4, 5]),
                   lambda w: write_residue(w, 2, 0, 1800, 32, 2, [3, 1, 4, 6], [3, 4, 5, 5, 4, 3])]
        two = name == "two_pass_slab"
        nch = 3 if name == "res0_3ch" else 2
        c.update(channels=nch, block0=256, block1=2048,
                 floors=[_floor1_small(0, 1), _floor1_long(0, 1, 10)],
                 residues=res,
                 mappings=[lambda w: write_mapping(w, nch, 2 if two else 1, [(0, 1)], [0, 1] if two else None, [(0, 0)] * (2 if two else 1)),
                           lambda w: write_mapping(w, nch, 2 if two else 1, [(nch - 1, 0)], [1, 0] if two else None, [(1, 1)] * (2 if two else 1))],
                 modes=[(0, 0), (1, 1)])
    elif name in ("table_books_pair", "table_books_general", "table_books_b1"):
        # books with an explicit table -- lookup type 2 (books 6, 9) and type 1 with sequence_p (book 7), Codebook.cs:262-281 -- in
        # residues the slab kernels take (round 6: kernels_synth.hip: table_value): on the pair walk (even dimensions, a per-channel
        # Residue1 and a stereo Residue2), on the general bin walk (dimensions 3 and 5 dividing 15 / 30), on the quirk-B-1 bin walk
        # (three channels, partitions off the channel grid); every residue mixes them with lattice books
        c["books"] = books + [
            Book(6, dims=2, lookup=2, min_me=(-9, -5), delta_me=(1, -4), value_bits=5),                       # 11: type 2, dim 2
            Book(7, dims=4, lookup=2, min_me=(-11, -6), delta_me=(1, -5), value_bits=5, sequence_p=1),       # 12: type 2 + sequence_p, dim 4
        ]
        if name == "table_books_pair":
            nch, cpl = 2, [(0, 1)]
            res = [lambda w: write_residue(w, 1, 0, 128, 16, 2, [1, 2, 7, 0], [7, 11, 3, 12, 7]),
                   lambda w: write_residue(w, 2, 0, 1800, 32, 2, [3, 1, 4, 6], [11, 4, 7, 12, 5, 11])]
        elif name == "table_books_general":
            nch, cpl = 2, [(0, 1)]
            res = [lambda w: write_residue(w, 1, 0, 120, 15, 2, [1, 2, 7, 0], [6, 9, 6, 9, 6]),
                   lambda w: write_residue(w, 2, 6, 1986, 30, 2, [3, 1, 4, 6], [6, 7, 9, 3, 6, 9])]
        else:
            nch, cpl = 3, [(0, 1), (2, 1)]
            res = [lambda w: write_residue(w, 2, 0, 360, 16, 2, [1, 3, 0, 2], [7, 11, 3, 12]),
                   lambda w: write_residue(w, 2, 4, 1500, 32, 2, [1, 3, 5, 2], [11, 7, 12, 3, 4, 7])]
        c.update(channels=nch, block0=256, block1=1024 if nch == 3 else 2048,
                 floors=[_floor1_small(0, 1), _floor1_long(0, 1, 9 if nch == 3 else 10)],
                 residues=res,
                 mappings=[lambda w: write_mapping(w, nch, 1, cpl, None, [(0, 0)]),
                           lambda w: write_mapping(w, nch, 1, cpl[::-1] if nch == 3 else [(1, 0)], None, [(1, 1)])],
                 modes=[(0, 0), (1, 1)])
    elif name in ("ch4_res1", "ch5_res2", "ch7_res1", "ch8_res2"):   # the channel counts no other config has
        nch, rtype = int(name[2]), int(name[-1])
        couple = [(0, 1), (2, 3)] if nch == 4 else [(0, 2), (3, 4)] if nch == 5 else [(0, 1), (2, 3), (5, 6)] if nch == 7 else [(0, 7), (1, 6), (2, 5)]
        mult = nch if rtype == 2 else 1
        c.update(channels=nch, block0=256, block1=2048,
                 floors=[_floor1_small(0, 1), _floor1_long(0, 1, 10)],
                 residues=[lambda w: write_residue(w, rtype, 0, mult * 120, 16, 2, [1, 2, 7, 0], [3, 4, 3, 4, 5]),
                           lambda w: write_residue(w, rtype, 8 * mult, mult * 960, 32, 2, [3, 1, 4, 6], [3, 4, 5, 5, 4, 3])],
                 mappings=[lambda w: write_mapping(w, nch, 1, couple, None, [(0, 0)]),
                           lambda w: write_mapping(w, nch, 1, couple[::-1], None, [(1, 1)])],
                 modes=[(0, 0), (1, 1)])
    else:
        raise KeyError(name)
    return c


CONFIG_NAMES = ["mono_res0_small_blocks", "stereo_res1_coupled", "three_ch_res2_misaligned", "six_ch_res2_4096",
                "floor0_stereo", "floor0_slab", "two_submaps", "equal_blocks_overrun", "mono_8192", "stereo_8192", "ch4_res1", "ch5_res2", "ch7_res1", "ch8_res2", "mono_res1_2048",
                "res0_slab", "odd_dims_slab", "res2_alias_stereo", "two_pass_slab", "res0_3ch",
                "table_books_pair", "table_books_general", "table_books_b1"]


def filtered_stream(oracle, name, npackets, seed, consistent_windows=True):
    """make_stream, with every audio packet that makes the reference algorithm throw (a floor curve leaving the
    dB table) replaced by a fresh random one, so that the stream as a whole decodes."""
    import ctypes as C
    cfg = config(name)
    packets, gr, fl = make_stream(cfg, npackets, seed, consistent_windows)
    hdr = packets[:3]
    blob = b"".join(hdr)
    offs = np.zeros(4, np.int64)
    offs[1:] = np.cumsum([len(p) for p in hdr])
    g3 = np.full(3, -1, np.int64)
    f3 = np.zeros(3, np.uint8)
    bb = np.frombuffer(blob, dtype=np.uint8)
    err = C.c_int(0)
    d = oracle.L.orc_open_packets(bb.ctypes.data, offs.ctypes.data, g3.ctypes.data, f3.ctypes.data, 3, C.byref(err))
    assert d, "synthetic setup header rejected: %d" % err.value
    try:
        ch, b1 = oracle.L.orc_channels(d), oracle.L.orc_block1(d)
        planes = np.zeros(ch * b1, np.float32)
        a, b, c, e = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rng = np.random.default_rng(seed + 99991)
        replaced = 0
        for i in range(3, len(packets)):
            for attempt in range(200):
                rc = oracle.L.orc_decode_packet_block(d, packets[i], len(packets[i]), planes.ctypes.data, C.byref(a), C.byref(b),
                                                      C.byref(c), C.byref(e))
                if rc >= 0 and np.isfinite(planes).all():
                    break
                head = packets[i][:1]
                body = rng.integers(0, 256, len(packets[i]) - 1).astype(np.uint8).tobytes()
                packets[i] = head + body
                replaced += 1
            else:
                raise AssertionError("could not find a decodable packet for " + name)
    finally:
        oracle.L.orc_close(d)
    return packets, gr, fl
