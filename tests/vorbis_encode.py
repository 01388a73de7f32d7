"""Structured Vorbis packet writer for parity tests (test infrastructure): the inverse of the bit parser.

Given a parsed setup (tests/vorbis_spec.py: written from the Vorbis I specification), PacketEncoder writes audio
packets whose side information is *chosen*, not random bits: floor posts, a classification for every residue partition
and a VQ entry for every vector of every cascade stage, each Huffman-encoded with the setup's own codewords.  Such
packets run the residue decode to its full depth -- every partition, all cascade stages -- which random-byte packets
never do (a real 6-channel n=4096 packet is 1-3 KB).

The bit order follows what the *reference* reads (Mapping.cs:95-133, Floor1.cs:135-184, Residue0.cs:119-178,
Residue2.cs:16-47), including where it departs from the specification: Residue0/1 read classwords and vectors for all
of the stream's channels whenever any channel is live (quirk B-2).

Builders at the bottom derive the BASELINE.json workload setups (C2 G-rand, C3 Markov, C4 six-channel with partition
size 48 / 32, the C5 corpus) from the headers of a shipped file.
"""
import numpy as np

from tests import ogg_py, vorbis_spec
from tests.synth_stream import BitWriter, comment_header, id_header, ilog, write_floor1, write_mapping, write_residue


def _bitrev(c, n):
    return vorbis_spec.bitrev(c, n)


class BookEnc:
    """Write side of one codebook: entry -> (bit-reversed codeword, length)."""

    def __init__(self, book):
        self.book = book
        self.code = {}
        for e, w in enumerate(book.words):
            if w is not None:
                self.code[e] = (_bitrev(w[0], w[1]), w[1])
        self.used = np.asarray(sorted(self.code), dtype=np.int64)

    def put(self, w, e):
        c, n = self.code[int(e)]
        w.write(c, n)


class PacketEncoder:
    def __init__(self, setup):
        self.S = setup
        self.enc = [BookEnc(b) for b in setup.books]
        self._floor_choices = {}

    # ---- floor 1 -----------------------------------------------------------------------------------
    def random_floor(self, rng, fl, y01=(16, 80), p_zero=0.5, geo=0.35):
        """Raw Y codes for one channel: two absolute posts, then 'geometric-ish' coded values (P(0) = p_zero), each
        reduced to something the partition's subclass books can encode.  Returns (ys, subclass choice per partition)."""
        S = self.S
        ys = [int(rng.integers(y01[0], min(y01[1], fl.range))), int(rng.integers(y01[0], min(y01[1], fl.range)))]
        subs = []
        for c in fl.partition_class:
            cdim, cbits = fl.class_dims[c], fl.class_subs[c]
            master = self.enc[fl.class_master[c]] if cbits else None
            for _attempt in range(64):
                vals, sel = [], []
                for _ in range(cdim):
                    want = 0 if rng.random() < p_zero else int(rng.geometric(geo))
                    options = []
                    for s_idx, bk in enumerate(fl.sub_books[c]):
                        if bk < 0:
                            options.append((s_idx, 0))
                        else:
                            used = self.enc[bk].used
                            ok = used[used <= want]
                            if ok.size:
                                options.append((s_idx, int(ok[-1])))
                    # prefer the option that keeps the wanted value
                    best = max(v for _, v in options)
                    pick = [o for o in options if o[1] == best]
                    s_idx, v = pick[int(rng.integers(0, len(pick)))]
                    vals.append(v)
                    sel.append(s_idx)
                cval = 0
                for j, s_idx in enumerate(sel):
                    cval |= s_idx << (cbits * j)
                if master is None or cval in master.code:
                    break
            else:
                raise AssertionError("no encodable subclass combination")
            ys.extend(vals)
            subs.append(cval)
        return ys, subs

    def put_floor(self, w, fl, ys, subs):
        if ys is None:
            w.write(0, 1)
            return
        w.write(1, 1)
        w.write(ys[0], fl.ybits)
        w.write(ys[1], fl.ybits)
        k = 2
        for c, cval in zip(fl.partition_class, subs):
            cdim, cbits = fl.class_dims[c], fl.class_subs[c]
            if cbits:
                self.enc[fl.class_master[c]].put(w, cval)
            csub = (1 << cbits) - 1
            for _ in range(cdim):
                bk = fl.sub_books[c][cval & csub]
                cval >>= cbits
                if bk >= 0:
                    self.enc[bk].put(w, ys[k])
                else:
                    assert ys[k] == 0
                k += 1

    # ---- residue -----------------------------------------------------------------------------------
    def put_residue(self, w, rng, res, block_size, class_weights=None, stats=None):
        """Residue0.Decode's read order (Residue0.cs:119-178) with random classifications / entries."""
        S = self.S
        ch = 1 if res.type == 2 else S.channels
        bs = block_size * S.channels if res.type == 2 else block_size
        end = min(res.end, bs // 2)
        n = end - res.begin
        if n <= 0:
            return
        nparts = n // res.psize
        cb = self.enc[res.classbook]
        cdim = cb.book.dims
        nclass = res.nclass
        max_stages = max(ilog(c) for c in res.cascade)
        words = (nparts + cdim - 1) // cdim
        # classification per (channel, partition): uniform over the classes (or weighted), classword must be a used entry
        cls = np.zeros((ch, words * cdim), dtype=np.int64)
        cw = np.zeros((ch, words), dtype=np.int64)
        p = None
        if class_weights is not None:
            p = np.asarray(class_weights, dtype=np.float64)
            p = p / p.sum()
        for c in range(ch):
            for k in range(words):
                for _ in range(256):
                    digits = rng.choice(nclass, size=cdim, p=p) if p is not None else rng.integers(0, nclass, cdim)
                    idx = 0
                    for d in digits:  # first partition = most significant digit (Residue0.cs:101-114)
                        idx = idx * nclass + int(d)
                    if idx in cb.code:
                        break
                else:
                    raise AssertionError("no usable classword")
                cw[c, k] = idx
                cls[c, k * cdim:(k + 1) * cdim] = digits
        for stage in range(max_stages):
            part = 0
            k = 0
            while part < nparts:
                if stage == 0:
                    for c in range(ch):
                        cb.put(w, cw[c, k])
                for _d in range(cdim):
                    if part >= nparts:
                        break
                    for c in range(ch):
                        idx = int(cls[c, part])
                        if res.cascade[idx] & (1 << stage):
                            bk = res.books[idx][stage]
                            if bk >= 0:
                                be = self.enc[bk]
                                dims = be.book.dims
                                # Residue0: psize / dims entries; Residue1/2: entries until psize values are written
                                cnt = res.psize // dims if res.type == 0 else (res.psize + dims - 1) // dims
                                es = be.used[rng.integers(0, be.used.size, cnt)]
                                for e in es:
                                    c_, n_ = be.code[int(e)]
                                    w.write(c_, n_)
                                if stats is not None:
                                    stats["vectors"] = stats.get("vectors", 0) + cnt
                                    stats.setdefault("stages", set()).add(stage)
                    part += 1
                k += 1

    # ---- whole packet ------------------------------------------------------------------------------
    def packet(self, rng, mode, prev_flag=1, next_flag=1, silent=(), floor_kw=None, class_weights=None, stats=None):
        """One audio packet of mode `mode`; channels listed in `silent` get an unused floor."""
        S = self.S
        w = BitWriter()
        w.write(0, 1)
        w.write(mode, S.mode_bits)
        long_block, mapping_idx = S.modes[mode]
        n = S.block1 if long_block else S.block0
        if long_block:
            w.write(prev_flag, 1)
            w.write(next_flag, 1)
        m = S.mappings[mapping_idx]
        for c in range(S.channels):
            fl = S.floors[m.submap_floor[m.mux[c]]]
            if c in silent:
                self.put_floor(w, fl, None, None)
            else:
                ys, subs = self.random_floor(rng, fl, **(floor_kw or {}))
                self.put_floor(w, fl, ys, subs)
        if len(silent) < S.channels:  # Residue0.cs:125: nothing is read when every channel is silent
            for sm in range(m.submaps):
                self.put_residue(w, rng, S.residues[m.submap_residue[sm]], n, class_weights, stats)
        return w.bytes()


# ---- block sequences, granules ---------------------------------------------------------------------------------------

def markov_kinds(rng, nframes, p_ls=0.03, p_sl=0.12, start_long=True):
    """block-kind sequence from a 2-state Markov chain; True = long."""
    kinds = np.zeros(nframes, dtype=bool)
    cur = start_long
    for i in range(nframes):
        kinds[i] = cur
        cur = (rng.random() >= p_ls) if cur else (rng.random() < p_sl)
    return kinds


def granules_for(setup, kinds):
    """Absolute sample position after each audio packet (spec 4.3.8: prev/4 + cur/4 per packet, none for the first)."""
    pos = 0
    out = []
    prev = None
    for k in kinds:
        n = setup.block1 if k else setup.block0
        if prev is not None:
            pos += prev // 4 + n // 4
        prev = n
        out.append(pos)
    return out


def encode_stream(setup, headers, kinds, seed, long_mode=None, short_mode=None, **kw):
    """Full-depth audio packets for a block-kind sequence with consistent window flags.  Returns (packets, granules)."""
    rng = np.random.default_rng(seed)
    enc = PacketEncoder(setup)
    if long_mode is None:
        long_mode = next(i for i, (f, _) in enumerate(setup.modes) if f)
    if short_mode is None:
        short_mode = next((i for i, (f, _) in enumerate(setup.modes) if not f), long_mode)
    packets = list(headers)
    n = len(kinds)
    for i in range(n):
        if kinds[i]:
            prev_flag = 1 if (i == 0 or kinds[i - 1]) else 0
            next_flag = 1 if (i + 1 >= n or kinds[i + 1]) else 0
            packets.append(enc.packet(rng, long_mode, prev_flag, next_flag, **kw))
        else:
            packets.append(enc.packet(rng, short_mode, **kw))
    gr = [-1, -1, -1] + granules_for(setup, kinds)
    return packets, gr


# ---- setups derived from a shipped file ----------------------------------------------------------------------

def shipped_headers(ogg_bytes):
    pk, _, _ = ogg_py.read_packets(ogg_bytes)
    return pk[:3]


def setup_of(headers):
    return vorbis_spec.Setup(headers[0], headers[2])


def c4_headers(base_headers, psize=48, channels=6, block0=256, block1=4096, rate=48000, end_per_channel=1536):
    """BASELINE C4: 6 channels, 48 kHz, blocks 256/4096, one submap, coupling [(0,2),(3,4)], Residue2 over all channels
    with end = 6*1536 and partition size 48 (headline) or 32 (quirk B-1), codebooks / residue books / floor classes
    taken from the base file (3test.ogg), long floor X list rescaled to rangebits 11."""
    base = setup_of(base_headers)
    long_mode = next(i for i, (f, _) in enumerate(base.modes) if f)
    short_mode = next(i for i, (f, _) in enumerate(base.modes) if not f)
    bm_long = base.mappings[base.modes[long_mode][1]]
    bm_short = base.mappings[base.modes[short_mode][1]]
    fl_long, fl_short = base.floors[bm_long.submap_floor[0]], base.floors[bm_short.submap_floor[0]]
    rs_long, rs_short = base.residues[bm_long.submap_residue[0]], base.residues[bm_short.submap_residue[0]]
    w = BitWriter()
    for b in b"\x05vorbis":
        w.write(b, 8)
    base.copy_book_bits(w)
    w.write(0, 6)
    w.write(0, 16)

    def put_floor(fl, rangebits):
        scale_from = ilog(fl.xs[1]) - 1
        xs = [x << (rangebits - scale_from) if rangebits >= scale_from else x >> (scale_from - rangebits) for x in fl.xs[2:]]
        ncls = max(fl.partition_class) + 1
        write_floor1(w, fl.partition_class, {c: fl.class_dims[c] for c in range(ncls)}, {c: fl.class_subs[c] for c in range(ncls)},
                     {c: fl.class_master[c] for c in range(ncls)}, {c: fl.sub_books[c] for c in range(ncls)}, fl.multiplier,
                     rangebits, xs)

    def put_res(rs, begin, end, ps):
        books = [b for row in rs.books for b in row if b >= 0]
        write_residue(w, 2, begin, end, ps, rs.classbook, rs.cascade, books)

    w.write(1, 6)  # two floors
    put_floor(fl_short, ilog(block0 // 2) - 1)
    put_floor(fl_long, ilog(block1 // 2) - 1)
    w.write(1, 6)  # two residues
    short_ps = 24 if psize % channels == 0 else 16
    put_res(rs_short, 0, channels * (block0 // 2) * 3 // 4 // short_ps * short_ps, short_ps)
    put_res(rs_long, 0, channels * end_per_channel, psize)
    w.write(1, 6)  # two mappings
    coupling = [(0, 2), (3, 4)] if channels >= 5 else ([(0, 1)] if channels >= 2 else [])
    write_mapping(w, channels, 1, coupling, None, [(0, 0)])
    write_mapping(w, channels, 1, coupling, None, [(1, 1)])
    w.write(1, 6)  # two modes
    for flag, mp in ((0, 0), (1, 1)):
        w.write(flag, 1)
        w.write(0, 16)
        w.write(0, 16)
        w.write(mp, 8)
    w.write(1, 1)
    return [id_header(channels, rate, block0, block1), comment_header(), w.bytes()]


def packet_pool(setup, seed, per_kind=64, **kw):
    """A pool of full-depth packets for every block kind (S, and L with each prev/next flag pair)."""
    rng = np.random.default_rng(seed)
    enc = PacketEncoder(setup)
    long_mode = next(i for i, (f, _) in enumerate(setup.modes) if f)
    short_mode = next((i for i, (f, _) in enumerate(setup.modes) if not f), None)
    pool = {}
    for pf in (0, 1):
        for nf in (0, 1):
            pool[(True, pf, nf)] = [enc.packet(rng, long_mode, pf, nf, **kw) for _ in range(per_kind if pf and nf else max(4, per_kind // 8))]
    if short_mode is not None:
        pool[(False, 1, 1)] = [enc.packet(rng, short_mode, **kw) for _ in range(per_kind)]
    return pool


def stream_from_pool(setup, headers, pool, kinds, rng):
    """A stream assembled from pooled packets (cheap: no encoding), window flags consistent with `kinds`."""
    packets = list(headers)
    n = len(kinds)
    for i in range(n):
        if kinds[i]:
            key = (True, 1 if (i == 0 or kinds[i - 1]) else 0, 1 if (i + 1 >= n or kinds[i + 1]) else 0)
        else:
            key = (False, 1, 1)
        lst = pool[key]
        packets.append(lst[int(rng.integers(0, len(lst)))])
    return packets, [-1, -1, -1] + granules_for(setup, kinds)


def corpus_file(setup, headers, pool, index, seconds_range=(5.0, 300.0), scale=1.0, p_ls=0.03, p_sl=0.12):
    """File `index` of the C5 corpus: length log-uniform in seconds_range (x scale), seed = file index,
    block kinds from the C3 Markov chain, packets drawn from the pool, laced into CRC-valid pages."""
    rng = np.random.default_rng(index)
    lo, hi = seconds_range
    seconds = float(np.exp(rng.uniform(np.log(lo), np.log(hi)))) * scale
    nframes = max(4, int(seconds * setup.rate / (setup.block1 // 2)))
    has_short = any(not f for f, _ in setup.modes)
    kinds = markov_kinds(rng, nframes, p_ls, p_sl) if has_short else np.ones(nframes, dtype=bool)
    packets, gr = stream_from_pool(setup, headers, pool, kinds, rng)
    return ogg_py.write_ogg(packets, gr, serial=0x10000 + index, page_packets=int(rng.integers(8, 40)))
