"""Independent OpenEXR scanline decoder (Python + numpy + zlib) for the reference-held images.

Test infrastructure only. Covers what the files under the reference's `scenes/evaluation/references/` use: single-part
scanline images, HALF or FLOAT channels, compression NONE / ZIPS / ZIP (zlib + byte predictor + interleave) and PIZ
(bitmap LUT + canonical Huffman with run-length symbol + two-dimensional Haar-like wavelet). Written from the published
OpenEXR file-layout description ("OpenEXR File Layout", "Technical Introduction to OpenEXR": PIZ = wavelet transform followed
by Huffman coding of 16-bit words); no OpenEXR / tinyexr code is available in this image. The product's own reader
(`ignis_amd/csrc/host/exr.h`, C++) is a second implementation; `tests/test_exr.py` checks them against each other.
"""
import struct
import zlib

import numpy as np

_PIXEL_SIZE = {0: 4, 1: 2, 2: 4}  # UINT, HALF, FLOAT
_LINES = {0: 1, 1: 1, 2: 1, 3: 16, 4: 32}  # scanlines per chunk: NONE, RLE, ZIPS, ZIP, PIZ


def parse_header(b):
    magic, version = struct.unpack_from("<II", b, 0)
    if magic != 20000630:
        raise ValueError("not an OpenEXR file")
    if version & 0x200 or version & 0x800 or version & 0x1000:
        raise ValueError("tiled / deep / multi-part files are not supported")
    pos, attrs = 8, {}
    while True:
        e = b.index(b"\0", pos)
        name = b[pos:e].decode()
        pos = e + 1
        if not name:
            break
        e = b.index(b"\0", pos)
        typ = b[pos:e].decode()
        pos = e + 1
        n, = struct.unpack_from("<i", b, pos)
        pos += 4
        attrs[name] = (typ, b[pos:pos + n])
        pos += n
    chans, ch, q = [], attrs["channels"][1], 0
    while ch[q] != 0:
        e = ch.index(b"\0", q)
        nm = ch[q:e].decode()
        q = e + 1
        ptype, _plin, xs, ys = struct.unpack_from("<iB3xii", ch, q)
        q += 16
        if xs != 1 or ys != 1:
            raise ValueError("subsampled channels are not supported")
        chans.append((nm, ptype))
    return {"channels": chans, "compression": attrs["compression"][1][0],
            "dataWindow": struct.unpack("<4i", attrs["dataWindow"][1]), "lineOrder": attrs["lineOrder"][1][0], "attrs": attrs}, pos


def _undo_zip(raw):
    """zlib stream -> predictor -> de-interleave (OpenEXR ZIP / ZIPS)."""
    t = np.frombuffer(zlib.decompress(raw), np.uint8).astype(np.int64)
    t[1:] -= 128
    t = (np.cumsum(t) & 255).astype(np.uint8)  # d[i] = d[i-1] + d[i] - 128 (mod 256)
    n = t.size
    half = (n + 1) // 2
    out = np.empty(n, np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


class _Bits:
    """MSB-first bit reader over bytes."""

    def __init__(self, data, pos=0):
        self.d, self.p, self.c, self.lc = data, pos, 0, 0

    def get(self, n):
        while self.lc < n:
            self.c = ((self.c << 8) | (self.d[self.p] if self.p < len(self.d) else 0)) & 0xFFFFFFFFFFFF
            self.p += 1
            self.lc += 8
        self.lc -= n
        return (self.c >> self.lc) & ((1 << n) - 1)


def _huf_uncompress(data, n_raw):
    """Canonical Huffman over 16-bit symbols with packed 6-bit code lengths and a run-length symbol (= iM)."""
    im, iM, _table_len, n_bits = struct.unpack_from("<IIII", data, 0)
    br = _Bits(data, 20)
    lengths = np.zeros(65537 + 300, np.int64)
    s = im
    while s <= iM:
        l = br.get(6)
        if l == 63:
            s += br.get(8) + 6
        elif l >= 59:
            s += l - 59 + 2
        else:
            lengths[s] = l
            s += 1
    lengths = lengths[:65537]
    # canonical codes: per length, codes are handed out in symbol order; start values from the longest length down
    n = np.bincount(lengths, minlength=59).astype(np.int64)
    start = np.zeros(59, np.int64)
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        start[i] = c
        c = nc
    table = {}
    nxt = start.copy()
    for sym in np.nonzero(lengths)[0]:
        l = int(lengths[sym])
        table[(l, int(nxt[l]))] = int(sym)
        nxt[l] += 1
    # fast path: every code of at most FAST bits fills a direct table indexed by the next FAST bits
    FAST = 12
    fast_sym = np.full(1 << FAST, -1, np.int64)
    fast_len = np.zeros(1 << FAST, np.int64)
    for (l, code), sym in table.items():
        if l <= FAST:
            lo = code << (FAST - l)
            fast_sym[lo:lo + (1 << (FAST - l))] = sym
            fast_len[lo:lo + (1 << (FAST - l))] = l
    fast_sym, fast_len = fast_sym.tolist(), fast_len.tolist()
    max_len = max(l for l, _ in table) if table else 0

    d, p = data, br.p  # the code table ends on a byte boundary of the reader
    end_bits = n_bits
    out = np.empty(n_raw, np.uint16)
    o = 0
    acc, nacc, used = 0, 0, 0
    nd = len(d)
    while o < n_raw:
        while nacc < 32 and p < nd:
            acc = (acc << 8) | d[p]
            p += 1
            nacc += 8
        if nacc < FAST:
            acc <<= (FAST - nacc)
            nacc_eff = FAST
        else:
            nacc_eff = nacc
        idx = (acc >> (nacc_eff - FAST)) & ((1 << FAST) - 1)
        sym, l = fast_sym[idx], fast_len[idx]
        if nacc_eff != nacc:
            acc >>= (FAST - nacc)
        if sym < 0:
            # long code: extend bit by bit
            l = FAST
            while True:
                l += 1
                if l > max_len or l > nacc:
                    raise ValueError("bad Huffman code")
                code = (acc >> (nacc - l)) & ((1 << l) - 1)
                if (l, code) in table:
                    sym = table[(l, code)]
                    break
        nacc -= l
        used += l
        acc &= (1 << nacc) - 1
        if sym == iM:
            while nacc < 8 and p < nd:
                acc = (acc << 8) | d[p]
                p += 1
                nacc += 8
            rep = (acc >> (nacc - 8)) & 255
            nacc -= 8
            used += 8
            acc &= (1 << nacc) - 1
            if o == 0 or o + rep > n_raw:
                raise ValueError("bad run in Huffman data")
            out[o:o + rep] = out[o - 1]
            o += rep
        else:
            out[o] = sym
            o += 1
        if used > end_bits:
            raise ValueError("Huffman data overrun")
    return out


def _wdec14(l, h):
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    ai = ls + (hs & 1) + (hs >> 1)
    return ai.astype(np.int16).astype(np.uint16), (ai - hs).astype(np.int16).astype(np.uint16)


def _wdec16(l, h):
    m, d = l.astype(np.int32), h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xFFFF
    aa = (d + bb - 0x8000) & 0xFFFF
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(a, max_value):
    """In-place inverse wavelet on a 2-D uint16 array [ny][nx] (one 16-bit component of one channel)."""
    ny, nx = a.shape
    dec = _wdec14 if max_value < (1 << 14) else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2)
        xs = np.arange(0, nx - p2 + 1, p2)
        if ys.size and xs.size:
            Y, X = np.meshgrid(ys, xs, indexing="ij")
            v00, v10, v01, v11 = a[Y, X], a[Y + p, X], a[Y, X + p], a[Y + p, X + p]
            i00, i10 = dec(v00, v10)
            i01, i11 = dec(v01, v11)
            r00, r01 = dec(i00, i01)
            r10, r11 = dec(i10, i11)
            a[Y, X], a[Y, X + p], a[Y + p, X], a[Y + p, X + p] = r00, r01, r10, r11
        if nx & p and ys.size:  # one more column without a horizontal partner
            x = xs[-1] + p2 if xs.size else 0
            lo, hi = dec(a[ys, x], a[ys + p, x])
            a[ys, x], a[ys + p, x] = lo, hi
        if ny & p and xs.size:  # one more row without a vertical partner
            y = ys[-1] + p2 if ys.size else 0
            lo, hi = dec(a[y, xs], a[y, xs + p])
            a[y, xs], a[y, xs + p] = lo, hi
        p2 = p
        p >>= 1


def _undo_piz(raw, chans, nx, ny):
    sizes = [_PIXEL_SIZE[t] // 2 for _, t in chans]  # 16-bit words per pixel
    n_words = sum(sizes) * nx * ny
    lo, hi = struct.unpack_from("<HH", raw, 0)
    pos = 4
    bitmap = np.zeros(8192, np.uint8)
    if lo <= hi:
        bitmap[lo:hi + 1] = np.frombuffer(raw, np.uint8, hi - lo + 1, pos)
        pos += hi - lo + 1
    bits = np.unpackbits(bitmap, bitorder="little").astype(bool)
    bits[0] = True  # zero is always part of the table
    lut = np.zeros(65536, np.uint16)
    present = np.nonzero(bits)[0]
    lut[:present.size] = present
    max_value = present.size - 1
    length, = struct.unpack_from("<i", raw, pos)
    pos += 4
    words = _huf_uncompress(raw[pos:pos + length], n_words)
    out = np.empty((ny, sum(sizes) * nx), np.uint16)
    start, col = 0, 0
    for size in sizes:
        block = words[start:start + nx * ny * size].reshape(ny, nx, size).copy()
        for j in range(size):
            comp = np.ascontiguousarray(block[:, :, j])
            _wav2_decode(comp, max_value)
            block[:, :, j] = comp
        out[:, col:col + nx * size] = lut[block.reshape(ny, nx * size)]
        start += nx * ny * size
        col += nx * size
    return out.astype("<u2").tobytes()


def read_exr(path):
    """-> dict channel name -> float32 array [height][width]."""
    b = open(path, "rb").read()
    h, pos = parse_header(b)
    x0, y0, x1, y1 = h["dataWindow"]
    W, H = x1 - x0 + 1, y1 - y0 + 1
    comp = h["compression"]
    if comp not in (0, 2, 3, 4):
        raise ValueError(f"compression {comp} is not supported")
    lines = _LINES[comp]
    chans = h["channels"]
    row_bytes = sum(_PIXEL_SIZE[t] for _, t in chans) * W
    n_chunks = (H + lines - 1) // lines
    offsets = struct.unpack_from(f"<{n_chunks}Q", b, pos)
    out = {nm: np.zeros((H, W), np.float32) for nm, _ in chans}
    for off in offsets:
        y, size = struct.unpack_from("<ii", b, off)
        raw = b[off + 8:off + 8 + size]
        ny = min(lines, y1 - y + 1)
        want = row_bytes * ny
        if size == want:
            data = raw  # stored uncompressed when compression would not have helped
        elif comp in (2, 3):
            data = _undo_zip(raw)
        elif comp == 4:
            data = _undo_piz(raw, chans, W, ny)
        else:
            raise ValueError("chunk size does not match an uncompressed image")
        if len(data) != want:
            raise ValueError("decoded chunk has the wrong size")
        q = 0
        for r in range(ny):
            for nm, t in chans:
                n = _PIXEL_SIZE[t] * W
                dt = {0: "<u4", 1: "<f2", 2: "<f4"}[t]
                out[nm][y - y0 + r] = np.frombuffer(data, dt, W, q).astype(np.float32)
                q += n
    return out


def read_rgb(path):
    c = read_exr(path)
    return np.stack([c["R"], c["G"], c["B"]], axis=-1)
