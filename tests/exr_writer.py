"""A small OpenEXR ENCODER, for tests only: writes the files that tests/test_exr_decode.py feeds to the C++ decoder
(include/trayracer_image.hpp).  Written from the published format documents independently of the decoder (Python, forward
transforms), because the reference's own EXR files are git-LFS stubs and no EXR library exists in the image -- so this pins
the decoder against a second reading of the same specification, not against another implementation ("parity unpinned").

Supports: single-part scan-line files and one-level tiled files; HALF / FLOAT / UINT channels; compression NONE (0), RLE (1),
ZIPS (2), ZIP (3), PIZ (4), PXR24 (5)."""
import heapq
import struct
import zlib

import numpy as np

NONE, RLE, ZIPS, ZIP, PIZ, PXR24 = 0, 1, 2, 3, 4, 5
UINT, HALF, FLOAT = 0, 1, 2
LINES = {NONE: 1, RLE: 1, ZIPS: 1, ZIP: 16, PIZ: 32, PXR24: 16}
STATS = {"compressed": 0, "raw": 0, "wavelet16": 0}   # blocks written since the last reset (tests check that a path was taken)


def _attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def _channel_bytes(arr, typ):
    if typ == HALF:
        return arr.astype(np.float16).view(np.uint16).astype("<u2").tobytes()
    if typ == FLOAT:
        return arr.astype("<f4").tobytes()
    return arr.astype("<u4").tobytes()


def _raw_block(channels, x0, y0, nx, ny):
    """Uncompressed layout: line by line, inside a line channel by channel (alphabetical), little endian."""
    out = bytearray()
    for y in range(y0, y0 + ny):
        for name, typ, arr in channels:
            out += _channel_bytes(arr[y, x0:x0 + nx], typ)
    return bytes(out)


# ---- RLE / ZIP ---------------------------------------------------------------------------------------------------
def _reorder_and_predict(raw):
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) & 255
    return d.astype(np.uint8).tobytes()


def _rle(data):
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        j = i + 1
        while j < n and data[j] == data[i] and j - i < 128:
            j += 1
        if j - i >= 3:
            out += struct.pack("b", j - i - 1) + data[i:i + 1]
            i = j
        else:
            k = i
            while k < n and k - i < 127 and not (k + 2 < n and data[k] == data[k + 1] == data[k + 2]):
                k += 1
            k = max(k, i + 1)
            out += struct.pack("b", -(k - i)) + data[i:k]
            i = k
    return bytes(out)


# ---- PXR24 -----------------------------------------------------------------------------------------------------
def _float_to_f24(f):
    """float32 -> the 24-bit pattern PXR24 keeps (sign, exponent, 15 mantissa bits, rounded; finite inputs only)."""
    u = np.asarray(f, "<f4").view(np.uint32).astype(np.uint64)
    s, e, m = u & 0x80000000, u & 0x7F800000, u & 0x007FFFFF
    assert not np.any(e == 0x7F800000), "finite values only"
    i = ((e | m) + (m & 0x80)) >> 8
    i = np.where(i >= 0x7F8000, (e | m) >> 8, i)   # rounding must not reach infinity: truncate instead
    return ((s >> 8) | i).astype(np.uint32)


def _pxr24(channels, x0, y0, nx, ny):
    out = bytearray()
    for y in range(y0, y0 + ny):
        for name, typ, arr in channels:
            row = arr[y, x0:x0 + nx]
            if typ == HALF:
                v = row.astype(np.float16).view(np.uint16).astype(np.uint32)
                d = np.diff(np.concatenate([[0], v])) & 0xFFFF
                out += ((d >> 8) & 255).astype(np.uint8).tobytes() + (d & 255).astype(np.uint8).tobytes()
            elif typ == FLOAT:
                v = _float_to_f24(row).astype(np.int64)
                d = np.diff(np.concatenate([[0], v])) & 0xFFFFFF
                out += ((d >> 16) & 255).astype(np.uint8).tobytes() + ((d >> 8) & 255).astype(np.uint8).tobytes() + (d & 255).astype(np.uint8).tobytes()
            else:
                v = row.astype(np.int64)
                d = np.diff(np.concatenate([[0], v])) & 0xFFFFFFFF
                out += b"".join(((d >> s) & 255).astype(np.uint8).tobytes() for s in (24, 16, 8, 0))
    return zlib.compress(bytes(out))


def pxr24_expected(arr, typ):
    """What a FLOAT channel decodes to after PXR24's 24-bit rounding (HALF and UINT are lossless)."""
    if typ != FLOAT:
        return arr
    return (_float_to_f24(arr) << 8).astype("<u4").view("<f4").reshape(arr.shape)


# ---- PIZ -------------------------------------------------------------------------------------------------------
def _wenc14(a, b):
    a = a - 65536 if a >= 32768 else a
    b = b - 65536 if b >= 32768 else b
    m = (a + b) >> 1
    d = a - b
    return m & 0xFFFF, d & 0xFFFF


def _wenc16(a, b):
    ao = (a + 0x8000) & 0xFFFF
    m = (ao + b) >> 1
    d = ao - b
    if d < 0:
        m = (m + 0x8000) & 0xFFFF
    return m, d & 0xFFFF


def _wav2_encode(buf, base, nx, ox, ny, oy, mx):
    enc = _wenc14 if mx < (1 << 14) else _wenc16
    if enc is _wenc16:
        STATS["wavelet16"] += 1
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        py = base
        ey = base + oy * (ny - p2)
        while py <= ey:
            px = py
            ex = py + ox * (nx - p2)
            while px <= ex:
                p01, p10 = px + ox1, px + oy1
                p11 = p10 + ox1
                i00, i01 = enc(buf[px], buf[p01])
                i10, i11 = enc(buf[p10], buf[p11])
                buf[px], buf[p10] = enc(i00, i10)
                buf[p01], buf[p11] = enc(i01, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1
                i00, buf[p10] = enc(buf[px], buf[p10])
                buf[px] = i00
            py += oy2
        if ny & p:
            px = py
            ex = py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1
                i00, buf[p01] = enc(buf[px], buf[p01])
                buf[px] = i00
                px += ox2
        p, p2 = p2, p2 << 1


class _BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0
        self.total = 0

    def put(self, nbits, value):
        self.acc = (self.acc << nbits) | value
        self.n += nbits
        self.total += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 255)
        self.acc &= (1 << self.n) - 1

    def finish(self):
        if self.n:
            self.out.append((self.acc << (8 - self.n)) & 255)
            self.n = 0
        return bytes(self.out)


LENGTH_HOOK = None   # callable(dict symbol -> code length) -> dict; None = write honest tables


def _huf_compress(symbols):
    freq = {}
    for s in symbols:
        freq[s] = freq.get(s, 0) + 1
    im, iM = min(freq), max(freq) + 1
    freq[iM] = 1   # the run-length pseudo symbol
    # code lengths by the textbook construction (any prefix-free lengths <= 58 are legal: codes are canonical)
    length = {s: 0 for s in freq}
    if len(freq) == 1:
        length[im] = 1
    else:
        heap = [(f, i, [s]) for i, (s, f) in enumerate(sorted(freq.items()))]
        heapq.heapify(heap)
        tick = len(heap)
        while len(heap) > 1:
            f1, _, s1 = heapq.heappop(heap)
            f2, _, s2 = heapq.heappop(heap)
            for s in s1 + s2:
                length[s] += 1
            heapq.heappush(heap, (f1 + f2, tick, s1 + s2))
            tick += 1
    assert max(length.values()) <= 58
    if LENGTH_HOOK is not None:   # tests of the decoder's table checks: tamper with the code lengths before they are packed
        length = LENGTH_HOOK(dict(length))
    count = [0] * 59
    for l in length.values():
        count[l] += 1
    start = [0] * 59
    c = 0
    for l in range(58, 0, -1):
        nc = (c + count[l]) >> 1
        start[l] = c
        c = nc
    code = {}
    for s in sorted(length):
        l = length[s]
        if l:
            code[s] = (l, start[l])
            start[l] += 1
    # packed table of code lengths
    tw = _BitWriter()
    s = im
    while s <= iM:
        l = length.get(s, 0)
        if l == 0:
            run = 1
            while s + run <= iM and length.get(s + run, 0) == 0 and run < 6 + 255:
                run += 1
            if run >= 2:
                if run >= 6:
                    tw.put(6, 63)
                    tw.put(8, run - 6)
                else:
                    tw.put(6, 59 + run - 2)
                s += run
                continue
        tw.put(6, l)
        s += 1
    table = tw.finish()
    dw = _BitWriter()
    rl, rc = code[iM]

    def send(sym, run):
        l, cde = code[sym]
        if l + rl + 8 < l * run:
            dw.put(l, cde)
            dw.put(rl, rc)
            dw.put(8, run)
        else:
            for _ in range(run + 1):
                dw.put(l, cde)

    prev, run = symbols[0], 0
    for v in symbols[1:]:
        if v == prev and run < 255:
            run += 1
        else:
            send(prev, run)
            run = 0
        prev = v
    send(prev, run)
    data = dw.finish()
    return struct.pack("<IIIII", im, iM, len(table), dw.total, 0) + table + data


def _piz(channels, x0, y0, nx, ny):
    words = []
    spans = []
    for name, typ, arr in channels:
        size = 1 if typ == HALF else 2
        blk = np.frombuffer(b"".join(_channel_bytes(arr[y, x0:x0 + nx], typ) for y in range(y0, y0 + ny)), "<u2")
        spans.append((len(words), size))
        words.extend(int(v) for v in blk)
    present = sorted(set(words))
    bitmap = bytearray(8192)
    for v in present:
        if v:
            bitmap[v >> 3] |= 1 << (v & 7)
    nz = [i for i, b in enumerate(bitmap) if b]
    mn, mx = (nz[0], nz[-1]) if nz else (8191, 0)
    lut, k = {}, 0
    for v in range(65536):
        if v == 0 or bitmap[v >> 3] & (1 << (v & 7)):
            lut[v] = k
            k += 1
    max_value = k - 1
    buf = [lut[v] for v in words]
    for base, size in spans:
        for j in range(size):
            _wav2_encode(buf, base + j, nx, size, ny, nx * size, max_value)
    huf = _huf_compress(buf)
    out = struct.pack("<HH", mn, mx)
    if mn <= mx:
        out += bytes(bitmap[mn:mx + 1])
    return out + struct.pack("<i", len(huf)) + huf


def _compress(comp, channels, x0, y0, nx, ny):
    raw = _raw_block(channels, x0, y0, nx, ny)
    if comp == NONE:
        return raw
    if comp == RLE:
        c = _rle(_reorder_and_predict(raw))
    elif comp in (ZIPS, ZIP):
        c = zlib.compress(_reorder_and_predict(raw))
    elif comp == PIZ:
        c = _piz(channels, x0, y0, nx, ny)
    else:
        c = _pxr24(channels, x0, y0, nx, ny)
    if len(c) < len(raw):
        STATS["compressed"] += 1
        return c
    STATS["raw"] += 1   # a block that does not shrink is stored raw
    return raw


def write_exr(path, channels, compression=NONE, tile=None, data_window_origin=(0, 0), line_order=0):
    """channels: dict name -> (pixel type, 2-D array); tile: None or (tile_w, tile_h)."""
    chans = sorted(((n, t, np.asarray(a)) for n, (t, a) in channels.items()), key=lambda c: c[0])
    h, w = chans[0][2].shape
    ox, oy = data_window_origin
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", t, 0, 0, 0, 0, 1, 1) for n, t, _ in chans) + b"\0"
    box = struct.pack("<iiii", ox, oy, ox + w - 1, oy + h - 1)
    header = (_attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([compression])) +
              _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box) +
              _attr("lineOrder", "lineOrder", bytes([line_order])) + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
              _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)))
    if tile:
        header += _attr("tiles", "tiledesc", struct.pack("<IIB", tile[0], tile[1], 0))
    header += b"\0"
    magic = struct.pack("<II", 20000630, 2 | (0x200 if tile else 0))
    chunks = []
    if tile:
        tw, th = tile
        for ty in range((h + th - 1) // th):
            for tx in range((w + tw - 1) // tw):
                nx, ny = min(tw, w - tx * tw), min(th, h - ty * th)
                data = _compress(compression, chans, tx * tw, ty * th, nx, ny)
                chunks.append(struct.pack("<iiiii", tx, ty, 0, 0, len(data)) + data)
    else:
        lines = LINES[compression]
        for y0 in range(0, h, lines):
            ny = min(lines, h - y0)
            data = _compress(compression, chans, 0, y0, w, ny)
            chunks.append(struct.pack("<ii", y0 + oy, len(data)) + data)
    order = list(range(len(chunks)))
    if line_order == 1 and not tile:   # decreasing y: the blocks are stored bottom-up, the offset table stays top-down
        order.reverse()
    table_pos = len(magic) + len(header)
    pos = table_pos + 8 * len(chunks)
    offsets = [0] * len(chunks)
    body = bytearray()
    for i in order:
        offsets[i] = pos + len(body)
        body += chunks[i]
    with open(path, "wb") as f:
        f.write(magic + header + b"".join(struct.pack("<Q", o) for o in offsets) + bytes(body))
