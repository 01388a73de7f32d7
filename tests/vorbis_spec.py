"""A second, independent Vorbis I decoder -- written from the Vorbis I specification, in double precision.

Test infrastructure.  Nothing here follows the C# reference or oracle/: the bit reader, the Huffman codeword
assignment ("lowest valued available codeword of that length", spec 3.2.1), the VQ lookup unpack (3.2.1/3.3),
floor 1 (7.2), residue 0/1/2 (8.6), channel coupling (4.3.5), the IMDCT as the plain cosine sum and the window /
overlap rules (4.3.1, 4.3.8) are written from the specification's text.  The purpose is to pin the oracle from a
second direction: a misreading of the C# that oracle and product share would show up as a disagreement with this
file on the shipped .ogg files (tests/test_spec_pin.py: <= 1e-6 absolute).

It also provides what tests/vorbis_encode.py needs: parsed setup structures (with the bit extents of every header
section, so that sections can be copied verbatim into new setup headers) and the codeword tables.

Deliberately NOT mirrored here (spec behaviour kept): residue decode honours the per-channel "do not decode" flags,
floor curves are rendered toward the post and truncated, the final granule trims the output.
"""
import math

import numpy as np


class EndOfPacket(Exception):
    pass


class BitReader:
    """LSb-first bit reader (spec section 2)."""

    def __init__(self, data):
        self.v = int.from_bytes(data, "little")
        self.nbits = len(data) * 8
        self.pos = 0

    def read(self, n):
        if n == 0:
            return 0
        if self.pos + n > self.nbits:
            self.pos = self.nbits
            raise EndOfPacket()
        r = (self.v >> self.pos) & ((1 << n) - 1)
        self.pos += n
        return r

    def peek32(self):
        return (self.v >> self.pos) & 0xFFFFFFFF

    def left(self):
        return self.nbits - self.pos


def ilog(x):
    return int(x).bit_length() if x > 0 else 0


def float32_unpack(x):
    mantissa = x & 0x1FFFFF
    exponent = (x & 0x7FE00000) >> 21
    if x & 0x80000000:
        mantissa = -mantissa
    return math.ldexp(float(mantissa), exponent - 788)


def lookup1_values(entries, dims):
    """Greatest integer r with r**dims <= entries (spec 9.2.3), in exact integer arithmetic."""
    r = int(round(entries ** (1.0 / dims)))
    while r ** dims > entries:
        r -= 1
    while (r + 1) ** dims <= entries:
        r += 1
    return r


def bitrev(c, n):
    r = 0
    for _ in range(n):
        r = (r << 1) | (c & 1)
        c >>= 1
    return r


def assign_codewords(lengths):
    """Spec 3.2.1: entries in order, each takes the lowest valued (leftmost) free codeword of its length.

    Free space is kept as a list of free subtrees (prefix value, depth), leftmost first.  Returns a list of
    (codeword, length) with the codeword MSb-first (first bit read = most significant), None for unused entries.
    A single used entry is the special case the spec allows (one codeword of the stated length, value 0)."""
    free = [(0, 0)]
    out = []
    for L in lengths:
        if L <= 0:
            out.append(None)
            continue
        pick = None
        for k, (p, d) in enumerate(free):  # leftmost free subtree that can hold a length-L word
            if d <= L:
                pick = k
                break
        if pick is None:
            raise ValueError("over-specified Huffman tree")
        p, d = free.pop(pick)
        code = p << (L - d)
        out.append((code, L))
        # the right siblings along the all-zero path below (p, d) become free, deepest last => keep list left-to-right
        sib = [(((code >> (L - k)) | 1), k) for k in range(d + 1, L + 1)]
        # left-to-right order in the tree: deeper siblings are further left
        sib.reverse()
        free[pick:pick] = sib
    return out


class Codebook:
    def __init__(self):
        self.dims = 0
        self.entries = 0
        self.lengths = []
        self.lookup_type = 0
        self.words = []  # (code MSb-first, length) or None
        self.bit_begin = self.bit_end = 0  # extent inside the setup packet

    @staticmethod
    def parse(r):
        b = Codebook()
        b.bit_begin = r.pos
        if r.read(24) != 0x564342:
            raise ValueError("codebook sync")
        b.dims = r.read(16)
        b.entries = r.read(24)
        ordered = r.read(1)
        lengths = []
        if not ordered:
            sparse = r.read(1)
            for _ in range(b.entries):
                if sparse:
                    lengths.append(r.read(5) + 1 if r.read(1) else 0)
                else:
                    lengths.append(r.read(5) + 1)
        else:
            cur = r.read(5) + 1
            while len(lengths) < b.entries:
                num = r.read(ilog(b.entries - len(lengths)))
                if len(lengths) + num > b.entries:
                    raise ValueError("ordered codebook overruns")
                lengths.extend([cur] * num)
                cur += 1
        b.lengths = lengths
        b.lookup_type = r.read(4)
        if b.lookup_type > 2:
            raise ValueError("lookup type")
        if b.lookup_type:
            b.minimum = float32_unpack(r.read(32))
            b.delta = float32_unpack(r.read(32))
            b.value_bits = r.read(4) + 1
            b.sequence_p = r.read(1)
            nvals = lookup1_values(b.entries, b.dims) if b.lookup_type == 1 else b.entries * b.dims
            b.lookup_values = nvals
            b.multiplicands = [r.read(b.value_bits) for _ in range(nvals)]
        b.bit_end = r.pos
        b._finish()
        return b

    def _finish(self):
        used = [L for L in self.lengths if L > 0]
        if len(used) == 1:
            self.words = [((0, L) if L > 0 else None) for L in self.lengths]
        else:
            self.words = assign_codewords(self.lengths)
        # decode tables: per code length, bit-reversed codeword -> entry (the stream is LSb-first)
        tabs = {}
        for e, w in enumerate(self.words):
            if w is not None:
                tabs.setdefault(w[1], {})[bitrev(w[0], w[1])] = e
        self.decode_tabs = sorted((L, (1 << L) - 1, t) for L, t in tabs.items())
        self.used_entries = [e for e, w in enumerate(self.words) if w is not None]
        self._vq = {}

    def decode_scalar(self, r):
        v = r.peek32()
        left = r.left()
        for L, mask, t in self.decode_tabs:
            if L > left:
                break
            e = t.get(v & mask)
            if e is not None:
                r.pos += L
                return e
        r.pos = r.nbits
        raise EndOfPacket()

    def vector(self, e):
        """VQ vector of entry e (spec 3.2.1 'VQ lookup table vector representation'), doubles."""
        got = self._vq.get(e)
        if got is not None:
            return got
        out = []
        last = 0.0
        if self.lookup_type == 1:
            div = 1
            for _ in range(self.dims):
                off = (e // div) % self.lookup_values
                val = self.multiplicands[off] * self.delta + self.minimum + last
                if self.sequence_p:
                    last = val
                out.append(val)
                div *= self.lookup_values
        elif self.lookup_type == 2:
            off = e * self.dims
            for i in range(self.dims):
                val = self.multiplicands[off + i] * self.delta + self.minimum + last
                if self.sequence_p:
                    last = val
                out.append(val)
        else:
            raise ValueError("scalar book used as VQ book")
        self._vq[e] = out
        return out


class Floor1:
    type = 1

    @staticmethod
    def parse(r):
        f = Floor1()
        parts = r.read(5)
        f.partition_class = [r.read(4) for _ in range(parts)]
        nclass = max(f.partition_class) + 1 if parts else 0
        f.class_dims, f.class_subs, f.class_master, f.sub_books = [], [], [], []
        for _ in range(nclass):
            f.class_dims.append(r.read(3) + 1)
            sub = r.read(2)
            f.class_subs.append(sub)
            f.class_master.append(r.read(8) if sub else -1)
            f.sub_books.append([r.read(8) - 1 for _ in range(1 << sub)])
        f.multiplier = r.read(2) + 1
        f.rangebits = r.read(4)
        f.xs = [0, 1 << f.rangebits]
        for c in f.partition_class:
            for _ in range(f.class_dims[c]):
                f.xs.append(r.read(f.rangebits))
        f.range = [256, 128, 86, 64][f.multiplier - 1]
        f.ybits = ilog(f.range - 1)
        n = len(f.xs)
        f.low = [0] * n
        f.high = [0] * n
        for i in range(2, n):  # low_neighbor / high_neighbor (9.2.4, 9.2.5)
            lo = max((k for k in range(i) if f.xs[k] < f.xs[i]), key=lambda k: f.xs[k])
            hi = min((k for k in range(i) if f.xs[k] > f.xs[i]), key=lambda k: f.xs[k])
            f.low[i], f.high[i] = lo, hi
        f.order = sorted(range(n), key=lambda k: f.xs[k])
        return f

    def decode(self, r, books):
        """Packet decode (7.2.3): list of raw Y values, or None ('unused')."""
        try:
            if not r.read(1):
                return None
            ys = [r.read(self.ybits), r.read(self.ybits)]
            for c in self.partition_class:
                cdim, cbits = self.class_dims[c], self.class_subs[c]
                csub = (1 << cbits) - 1
                cval = books[self.class_master[c]].decode_scalar(r) if cbits else 0
                for _ in range(cdim):
                    bk = self.sub_books[c][cval & csub]
                    cval >>= cbits
                    ys.append(books[bk].decode_scalar(r) if bk >= 0 else 0)
            return ys
        except EndOfPacket:
            return None

    def unwrap(self, ys):
        """Amplitude value synthesis (7.2.4 step 1): (final_Y, step2_flag)."""
        n = len(self.xs)
        final = [0] * n
        flag = [False] * n
        final[0], final[1] = ys[0], ys[1]
        flag[0] = flag[1] = True
        for i in range(2, n):
            lo, hi = self.low[i], self.high[i]
            pred = render_point(self.xs[lo], final[lo], self.xs[hi], final[hi], self.xs[i])
            val = ys[i]
            highroom = self.range - pred
            lowroom = pred
            room = 2 * min(highroom, lowroom)
            if val:
                flag[lo] = flag[hi] = flag[i] = True
                if val >= room:
                    final[i] = val - lowroom + pred if highroom > lowroom else pred - val + highroom - 1
                else:
                    final[i] = pred - (val + 1) // 2 if val & 1 else pred + val // 2
            else:
                final[i] = pred
        return final, flag

    def curve(self, ys, n2):
        """Curve synthesis (7.2.4 step 2): n2 integer floor values (dB table indices)."""
        final, flag = self.unwrap(ys)
        out = [0] * max(n2, self.xs[1] + 1)
        hx = lx = 0
        ly = final[self.order[0]] * self.multiplier
        hy = ly
        for k in self.order[1:]:
            if flag[k]:
                hy = final[k] * self.multiplier
                hx = self.xs[k]
                render_line(lx, ly, hx, hy, out)
                lx, ly = hx, hy
        if hx < n2:
            render_line(hx, hy, n2, hy, out)
        return out[:n2]


def render_point(x0, y0, x1, y1, x):
    dy = y1 - y0
    adx = x1 - x0
    off = abs(dy) * (x - x0) // adx
    return y0 - off if dy < 0 else y0 + off


def render_line(x0, y0, x1, y1, v):
    dy = y1 - y0
    adx = x1 - x0
    if adx <= 0:
        return
    base = int(dy / adx)  # toward zero
    sy = base - 1 if dy < 0 else base + 1
    ady = abs(dy) - abs(base) * adx
    y = y0
    err = 0
    if x0 < len(v):
        v[x0] = y
    for x in range(x0 + 1, min(x1, len(v))):
        err += ady
        if err >= adx:
            err -= adx
            y += sy
        else:
            y += base
        v[x] = y


def inverse_db(i):
    """floor1_inverse_dB_table[i] (spec 10.1) in closed form: a geometric ramp from 1.0649863e-07 to 1."""
    return math.exp(math.log(1.0649863e-07) * (255 - i) / 255.0)


class Residue:
    @staticmethod
    def parse(r, rtype):
        s = Residue()
        s.type = rtype
        s.begin = r.read(24)
        s.end = r.read(24)
        s.psize = r.read(24) + 1
        s.nclass = r.read(6) + 1
        s.classbook = r.read(8)
        s.cascade = []
        for _ in range(s.nclass):
            low = r.read(3)
            high = r.read(5) if r.read(1) else 0
            s.cascade.append(high * 8 + low)
        s.books = [[(r.read(8) if (c >> j) & 1 else -1) for j in range(8)] for c in s.cascade]
        return s


class Mapping:
    @staticmethod
    def parse(r, channels):
        m = Mapping()
        if r.read(16) != 0:
            raise ValueError("mapping type")
        m.submaps = r.read(4) + 1 if r.read(1) else 1
        m.coupling = []
        if r.read(1):
            steps = r.read(8) + 1
            bits = ilog(channels - 1)
            for _ in range(steps):
                mag = r.read(bits)
                ang = r.read(bits)
                m.coupling.append((mag, ang))
        if r.read(2) != 0:
            raise ValueError("mapping reserved bits")
        m.mux = [r.read(4) for _ in range(channels)] if m.submaps > 1 else [0] * channels
        m.submap_floor, m.submap_residue = [], []
        for _ in range(m.submaps):
            r.read(8)
            m.submap_floor.append(r.read(8))
            m.submap_residue.append(r.read(8))
        return m


class Setup:
    """Identification + setup header, parsed per spec 4.2.2 / 4.2.4."""

    def __init__(self, id_pkt, setup_pkt):
        r = BitReader(id_pkt)
        if bytes(r.read(8) for _ in range(7)) != b"\x01vorbis":
            raise ValueError("identification header")
        if r.read(32) != 0:
            raise ValueError("vorbis version")
        self.channels = r.read(8)
        self.rate = r.read(32)
        self.bitrates = (r.read(32), r.read(32), r.read(32))
        self.block0 = 1 << r.read(4)
        self.block1 = 1 << r.read(4)
        if not r.read(1):
            raise ValueError("framing")
        self.setup_bytes = bytes(setup_pkt)
        r = BitReader(setup_pkt)
        if bytes(r.read(8) for _ in range(7)) != b"\x05vorbis":
            raise ValueError("setup header")
        self.books_bit_begin = r.pos
        nbooks = r.read(8) + 1
        self.books = [Codebook.parse(r) for _ in range(nbooks)]
        self.books_bit_end = r.pos
        for _ in range(r.read(6) + 1):
            if r.read(16) != 0:
                raise ValueError("time domain transform")
        self.floors = []
        for _ in range(r.read(6) + 1):
            t = r.read(16)
            if t != 1:
                raise ValueError("only floor 1 is implemented in the spec-derived decoder (floor type %d)" % t)
            self.floors.append(Floor1.parse(r))
        self.residues = []
        for _ in range(r.read(6) + 1):
            t = r.read(16)
            if t > 2:
                raise ValueError("residue type")
            self.residues.append(Residue.parse(r, t))
        self.mappings = [Mapping.parse(r, self.channels) for _ in range(r.read(6) + 1)]
        self.modes = []
        for _ in range(r.read(6) + 1):
            flag = r.read(1)
            if r.read(16) != 0 or r.read(16) != 0:
                raise ValueError("mode window/transform type")
            self.modes.append((flag, r.read(8)))
        if not r.read(1):
            raise ValueError("framing")
        self.mode_bits = ilog(len(self.modes) - 1)
        self._windows = {}
        self._imdct = {}

    def copy_book_bits(self, w):
        """Append the codebook section of the setup header ('count - 1' byte + every book) verbatim to BitWriter w."""
        v = int.from_bytes(self.setup_bytes, "little")
        pos = self.books_bit_begin
        end = self.books_bit_end
        while pos < end:
            n = min(32, end - pos)
            w.write((v >> pos) & ((1 << n) - 1), n)
            pos += n

    # ---- window (4.3.1) and IMDCT (as the definition) ----
    def window(self, n, long_block, prev_flag, next_flag):
        key = (n, long_block, prev_flag, next_flag)
        w = self._windows.get(key)
        if w is not None:
            return w
        b0 = self.block0
        if long_block and not prev_flag:
            ls, le, ln = n // 4 - b0 // 4, n // 4 + b0 // 4, b0 // 2
        else:
            ls, le, ln = 0, n // 2, n // 2
        if long_block and not next_flag:
            rs, re, rn = n * 3 // 4 - b0 // 4, n * 3 // 4 + b0 // 4, b0 // 2
        else:
            rs, re, rn = n // 2, n, n // 2
        w = np.zeros(n)
        i = np.arange(ls, le)
        w[ls:le] = np.sin(np.pi / 2 * np.sin((i - ls + 0.5) / ln * np.pi / 2) ** 2)
        w[le:rs] = 1.0
        i = np.arange(rs, re)
        w[rs:re] = np.sin(np.pi / 2 * np.sin((i - rs + 0.5) / rn * np.pi / 2 + np.pi / 2) ** 2)
        self._windows[key] = w
        return w

    def imdct(self, X):
        """y[i] = sum_k X[k] cos(2 pi / n (i + 1/2 + n/4)(k + 1/2)), i < n (unnormalised, as Vorbis uses it)."""
        n2 = len(X)
        n = 2 * n2
        M = self._imdct.get(n)
        if M is None:
            i = np.arange(n)[:, None] + 0.5 + n / 4.0
            k = np.arange(n2)[None, :] + 0.5
            M = np.cos(2.0 * np.pi / n * i * k) if n <= 4096 else None
            self._imdct[n] = M
        if M is not None:
            return M @ X
        out = np.zeros(n)
        k = np.arange(n2) + 0.5
        for i0 in range(0, n, 256):
            i = np.arange(i0, i0 + 256)[:, None] + 0.5 + n / 4.0
            out[i0:i0 + 256] = np.cos(2.0 * np.pi / n * i * k[None, :]) @ X
        return out


def decode_residue(setup, res, r, do_not_decode, n2, nch):
    """Spec 8.6.2 .. 8.6.5.  Returns nch vectors of n2 doubles (the channels of this submap, in order)."""
    books = setup.books
    if res.type == 2:
        if all(do_not_decode):
            return [np.zeros(n2) for _ in range(nch)]
        vec = _decode_residue_01(setup, res, r, [False], n2 * nch, 1, 1)[0]
        return [vec[c::nch].copy() for c in range(nch)]
    return _decode_residue_01(setup, res, r, do_not_decode, n2, nch, res.type)


def _decode_residue_01(setup, res, r, do_not_decode, size, nch, fmt):
    books = setup.books
    out = [np.zeros(size) for _ in range(nch)]
    begin = min(res.begin, size)
    end = min(res.end, size)
    cb = books[res.classbook]
    cpc = cb.dims
    n_to_read = end - begin
    nparts = n_to_read // res.psize
    if n_to_read <= 0:
        return out
    cls = [[0] * (nparts + cpc) for _ in range(nch)]
    try:
        for stage in range(8):
            p = 0
            while p < nparts:
                if stage == 0:
                    for j in range(nch):
                        if not do_not_decode[j]:
                            temp = cb.decode_scalar(r)
                            for i in range(cpc - 1, -1, -1):
                                cls[j][i + p] = temp % res.nclass
                                temp //= res.nclass
                i = 0
                while i < cpc and p < nparts:
                    for j in range(nch):
                        if not do_not_decode[j]:
                            bk = res.books[cls[j][p]][stage]
                            if bk >= 0:
                                book = books[bk]
                                off = begin + p * res.psize
                                v = out[j]
                                if fmt == 0:
                                    step = res.psize // book.dims
                                    for a in range(step):
                                        vec = book.vector(book.decode_scalar(r))
                                        for d in range(book.dims):
                                            v[off + a + d * step] += vec[d]
                                else:
                                    a = 0
                                    while a < res.psize:
                                        vec = book.vector(book.decode_scalar(r))
                                        for d in range(book.dims):
                                            v[off + a] += vec[d]
                                            a += 1
                    p += 1
                    i += 1
    except EndOfPacket:
        pass
    return out


class SpecDecoder:
    """Audio packet decode (4.3) + overlap-add; feed packets in order with packet()."""

    def __init__(self, id_pkt, setup_pkt):
        self.setup = Setup(id_pkt, setup_pkt)
        self.prev_tail = None  # right half of the previous windowed block, per channel
        self.prev_n = 0
        self._db = np.array([inverse_db(i) for i in range(256)])

    def block(self, pkt):
        """One audio packet -> (n, windowed block [channels][n]) or None if the packet is to be discarded."""
        S = self.setup
        r = BitReader(pkt)
        try:
            if r.read(1) != 0:
                return None
            mode = r.read(S.mode_bits)
            long_block, mapping_idx = S.modes[mode]
            n = S.block1 if long_block else S.block0
            prev_flag = next_flag = 0
            if long_block:
                prev_flag = r.read(1)
                next_flag = r.read(1)
        except (EndOfPacket, IndexError):
            return None
        m = S.mappings[mapping_idx]
        n2 = n // 2
        ch = S.channels
        floors = []
        no_residue = []
        for c in range(ch):
            fl = S.floors[m.submap_floor[m.mux[c]]]
            ys = fl.decode(r, S.books)
            floors.append((fl, ys))
            no_residue.append(ys is None)
        for mag, ang in m.coupling:  # nonzero vector propagate (4.3.3)
            if not (no_residue[mag] and no_residue[ang]):
                no_residue[mag] = no_residue[ang] = False
        spectra = [None] * ch
        for sm in range(m.submaps):
            chans = [c for c in range(ch) if m.mux[c] == sm]
            res = S.residues[m.submap_residue[sm]]
            vecs = decode_residue(S, res, r, [no_residue[c] for c in chans], n2, len(chans))
            for c, v in zip(chans, vecs):
                spectra[c] = v
        for mag, ang in reversed(m.coupling):  # inverse coupling (4.3.5)
            M, A = spectra[mag], spectra[ang]
            newM = np.where(M > 0, np.where(A > 0, M, M + A), np.where(A > 0, M, M - A))
            newA = np.where(M > 0, np.where(A > 0, M - A, M), np.where(A > 0, M + A, M))
            spectra[mag], spectra[ang] = newM, newA
        w = S.window(n, bool(long_block), prev_flag, next_flag)
        out = np.zeros((ch, n))
        for c in range(ch):
            fl, ys = floors[c]
            if ys is None:
                continue  # unused floor: the channel's spectrum is zero (4.3.6)
            curve = self._db[np.asarray(fl.curve(ys, n2), dtype=np.int64)]
            out[c] = S.imdct(spectra[c] * curve) * w
        return n, out

    def packet(self, pkt):
        """Returns the samples this packet finishes, shape (channels, count); count 0 for the first packet."""
        got = self.block(pkt)
        ch = self.setup.channels
        if got is None:
            return np.zeros((ch, 0))
        n, blk = got
        if self.prev_tail is None:
            self.prev_tail = blk[:, n // 2:].copy()
            self.prev_n = n
            return np.zeros((ch, 0))
        pn = self.prev_n
        count = pn // 4 + n // 4
        out = np.zeros((ch, count))
        # output axis: index 0 = centre of the previous block; the current block starts at pn/4 - n/4
        off = pn // 4 - n // 4
        tl = self.prev_tail.shape[1]
        out[:, :min(tl, count)] += self.prev_tail[:, :min(tl, count)]
        lo = max(0, off)
        out[:, lo:count] += blk[:, lo - off:count - off]
        self.prev_tail = blk[:, n // 2:].copy()
        self.prev_n = n
        return out


def decode_ogg_packets(packets, final_granule=None):
    """packets: all packets of one logical stream (3 headers first).  Interleaved float64 PCM, spec semantics."""
    dec = SpecDecoder(packets[0], packets[2])
    outs = [dec.packet(p) for p in packets[3:]]
    pcm = np.concatenate(outs, axis=1) if outs else np.zeros((dec.setup.channels, 0))
    if final_granule is not None and final_granule >= 0:
        pcm = pcm[:, :final_granule]
    return pcm.T.reshape(-1).copy(), dec.setup
