"""pt_image_read_float (the environment source files the reference takes — .exr / .hdr, Rtxpt/Sample.cpp:116 — read without Donut): CPU only.
  * a third-party file: tests/golden/cpython_imghdr_python.exr (CPython's own imghdr test image: 16x16, half A B G R, uncompressed) against a numpy decode of its bytes;
  * files written here from the published layouts — OpenEXR scan-line with NONE / RLE / ZIPS / ZIP blocks (zlib + the byte predictor / de-interleave of ImfZip),
    half and float channels in any order, a data window that does not start at 0, luminance-only; Radiance RGBE flat and run-length;
  * what must be refused (PIZ, tiled, multi-part, sub-sampled channels) and 600 damaged files that must fail cleanly."""
import os, struct, sys, zlib
import numpy as np
import pytest

import rtxpt_amd as pt

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cpython_imghdr_python.exr")


def _attr(name, typ, val):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val


def _exr_transform(raw):
    """ImfZip / ImfRle: reorder into (even bytes, odd bytes), then delta-encode"""
    a = np.frombuffer(raw, np.uint8); t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
    d = t.copy(); d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
    return d.astype(np.uint8).tobytes()


def _exr_rle(data):
    out = bytearray(); i = 0; n = len(data)
    while i < n:
        j = i
        while j + 1 < n and data[j + 1] == data[i] and j - i < 126: j += 1
        if j - i >= 2:
            out += struct.pack("b", j - i) + bytes([data[i]]); i = j + 1
        else:
            k = i
            while k < n and k - i < 127 and not (k + 2 < n and data[k] == data[k + 1] == data[k + 2]): k += 1
            out += struct.pack("b", -(k - i)) + data[i:k]; i = k
    return bytes(out)


# ---- a PIZ encoder for the tests (OpenEXR's ImfPizCompressor / ImfHuf / ImfWav, encode side, written independently of the decoder in pt_hdrimage.cpp): value bitmap + forward
# lookup table, two-dimensional Haar wavelet per channel plane (14-bit or 16-bit modular arithmetic), Huffman coding with canonical codes and one run-length symbol
def _s16(v): v &= 0xFFFF; return v - 0x10000 if v & 0x8000 else v


def _wenc14(a, b):
    a_, b_ = _s16(a), _s16(b); return ((a_ + b_) >> 1) & 0xFFFF, (a_ - b_) & 0xFFFF


def _wenc16(a, b):
    ao = (a + 0x8000) & 0xFFFF; m = (ao + b) >> 1; d = ao - b
    if d < 0: m = (m + 0x8000) & 0xFFFF
    return m, d & 0xFFFF


def _wav2_encode(buf, base, nx, ox, ny, oy, mx):
    enc = _wenc14 if mx < (1 << 14) else _wenc16
    n = min(nx, ny); p, p2 = 1, 2
    while p2 <= n:
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        py = base; ey = base + oy * (ny - p2)
        while py <= ey:
            px = py; ex = py + ox * (nx - p2)
            while px <= ex:
                p01, p10 = px + ox1, px + oy1; p11 = p10 + ox1
                i00, i01 = enc(buf[px], buf[p01]); i10, i11 = enc(buf[p10], buf[p11])
                buf[px], buf[p10] = enc(i00, i10); buf[p01], buf[p11] = enc(i01, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1; buf[px], buf[p10] = enc(buf[px], buf[p10])
            py += oy2
        if ny & p:
            px = py; ex = py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1; buf[px], buf[p01] = enc(buf[px], buf[p01]); px += ox2
        p, p2 = p2, p2 << 1


class _Bits:
    def __init__(self): self.out = bytearray(); self.c = 0; self.lc = 0; self.n = 0

    def put(self, nbits, bits):
        self.c = (self.c << nbits) | bits; self.lc += nbits; self.n += nbits
        while self.lc >= 8: self.lc -= 8; self.out.append((self.c >> self.lc) & 0xFF)
        self.c &= (1 << self.lc) - 1

    def flush(self):
        if self.lc: self.out.append((self.c << (8 - self.lc)) & 0xFF); self.c = 0; self.lc = 0
        return bytes(self.out)


def _huf_compress(words, use_runs=True):
    import heapq
    freq = {}
    for v in words: freq[v] = freq.get(v, 0) + 1
    im = min(freq); iM = max(freq) + 1; freq[iM] = 1                      # the run-length symbol
    heap = [(f, i, (s,)) for i, (s, f) in enumerate(sorted(freq.items()))]; heapq.heapify(heap); length = {s: 0 for s in freq}; k = len(heap)
    if len(heap) == 1: length[heap[0][2][0]] = 1
    while len(heap) > 1:
        f1, _, s1 = heapq.heappop(heap); f2, _, s2 = heapq.heappop(heap)
        for s_ in s1 + s2: length[s_] += 1
        heapq.heappush(heap, (f1 + f2, k, s1 + s2)); k += 1
    assert max(length.values()) <= 58
    # canonical codes from the lengths alone (hufCanonicalCodeTable)
    n = [0] * 59
    for s_ in range(65537): n[length.get(s_, 0)] += 1
    c = 0
    for i in range(58, 0, -1): nc = (c + n[i]) >> 1; n[i] = c; c = nc
    code = {}
    for s_ in range(im, iM + 1):
        l = length.get(s_, 0)
        if l: code[s_] = n[l]; n[l] += 1
    tb = _Bits(); s_ = im                                                 # hufPackEncTable: 6-bit lengths, zero runs
    while s_ <= iM:
        l = length.get(s_, 0)
        if l == 0:
            z = 1
            while s_ + z <= iM and z < 255 + 6 and length.get(s_ + z, 0) == 0: z += 1
            if z >= 2:
                if z >= 6: tb.put(6, 63); tb.put(8, z - 6)
                else: tb.put(6, 59 + z - 2)
                s_ += z; continue
        tb.put(6, l); s_ += 1
    table = tb.flush()
    db = _Bits(); i = 0; nw = len(words)
    while i < nw:
        v = words[i]; r = 0
        while use_runs and i + r + 1 < nw and words[i + r + 1] == v and r < 255: r += 1
        if r and length[v] + length[iM] + 8 < length[v] * r:
            db.put(length[v], code[v]); db.put(length[iM], code[iM]); db.put(8, r)
        else:
            for _ in range(r + 1): db.put(length[v], code[v])
        i += r + 1
    nbits = db.n; data = db.flush()
    return struct.pack("<IIIII", im, iM, len(table), nbits, 0) + table + data


def _piz_block(lines, names, chans, use_runs=True):
    """lines: the y range of the block; returns the PIZ payload (or the raw bytes when PIZ does not shrink them)"""
    raw = b"".join(chans[n][y].tobytes() for y in lines for n in names)
    planes = []                                                            # channel after channel: [y][x][16-bit words of the pixel]
    for n in names: planes.append(np.concatenate([np.frombuffer(chans[n][y].tobytes(), np.uint16) for y in lines]))
    allw = np.concatenate(planes)
    present = np.zeros(65536, bool); present[allw] = True
    bitmap = np.packbits(present, bitorder="little")
    present[0] = True; fwd = np.cumsum(present) - 1; maxv = int(fwd[-1])   # zero is always in the table
    nz = np.nonzero(bitmap)[0]; lo, hi = (int(nz[0]), int(nz[-1])) if nz.size else (8191, 0)
    buf = [int(v) for v in fwd[allw]]; base = 0; w = chans[names[0]].shape[1]
    for n, pl in zip(names, planes):
        wk = 1 if chans[n].dtype == np.float16 else 2
        for j in range(wk): _wav2_encode(buf, base + j, w, wk, len(lines), w * wk, maxv)
        base += pl.size
    huf = _huf_compress(buf, use_runs)
    out = struct.pack("<HH", lo, hi) + (bitmap[lo:hi + 1].tobytes() if lo <= hi else b"") + struct.pack("<i", len(huf)) + huf
    return out if len(out) < len(raw) else raw


def write_exr(path, chans, comp, origin=(0, 0), version=2, extra_attrs=b""):
    """chans: {name: float32 or float16 array [h, w]}; comp: 0 none, 1 RLE, 2 ZIPS, 3 ZIP, 4 PIZ"""
    names = sorted(chans); h, w = chans[names[0]].shape
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 1 if chans[n].dtype == np.float16 else 2, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    x0, y0 = origin
    hdr = struct.pack("<ii", 20000630, version) + _attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([comp])) + \
        _attr("dataWindow", "box2i", struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)) + _attr("displayWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1)) + \
        _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + \
        _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + extra_attrs + b"\0"
    lpb = 16 if comp == 3 else (32 if comp == 4 else 1)
    blocks = []
    for yb in range(0, h, lpb):
        raw = b"".join(chans[n][y].tobytes() for y in range(yb, min(h, yb + lpb)) for n in names)
        if comp == 0: data = raw
        elif comp == 4: data = _piz_block(range(yb, min(h, yb + lpb)), names, chans)
        else:
            t = _exr_transform(raw); data = _exr_rle(t) if comp == 1 else zlib.compress(t)
            if len(data) >= len(raw): data = raw
        blocks.append(struct.pack("<ii", y0 + yb, len(data)) + data)
    off = len(hdr) + 8 * len(blocks); table = b""
    for b in blocks: table += struct.pack("<Q", off); off += len(b)
    open(path, "wb").write(hdr + table + b"".join(blocks))


def write_exr_tiled(path, chans, comp, tile, level_mode=0, origin=(0, 0)):
    """a single-part TILED file (version bit 0x200, attribute `tiles`): level (0, 0) holds the image; level_mode 1 (MIPMAP_LEVELS, rounding down) appends the further levels'
    tiles — filled with a constant — after it, in the offset table as in the file, so a reader that took anything but level 0 would show it."""
    names = sorted(chans); h, w = chans[names[0]].shape; tw, th = tile
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 1 if chans[n].dtype == np.float16 else 2, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    x0, y0 = origin
    hdr = struct.pack("<ii", 20000630, 2 | 0x200) + _attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([comp])) + \
        _attr("dataWindow", "box2i", struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)) + _attr("displayWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1)) + \
        _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + \
        _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + _attr("tiles", "tiledesc", struct.pack("<IIB", tw, th, level_mode)) + b"\0"
    levels = [dict(chans)]
    if level_mode == 1:
        lw, lh = w, h
        while lw > 1 or lh > 1:
            lw, lh = max(1, lw // 2), max(1, lh // 2)
            levels.append({n: np.full((lh, lw), 777.0, chans[n].dtype) for n in names})
    chunks = []
    for l, lc in enumerate(levels):
        lh, lw = lc[names[0]].shape
        for ty in range((lh + th - 1) // th):
            for tx in range((lw + tw - 1) // tw):
                sub = {n: np.ascontiguousarray(lc[n][ty * th:min(lh, (ty + 1) * th), tx * tw:min(lw, (tx + 1) * tw)]) for n in names}
                rows = sub[names[0]].shape[0]
                raw = b"".join(sub[n][y].tobytes() for y in range(rows) for n in names)
                if comp == 0: data = raw
                elif comp == 4: data = _piz_block(range(rows), names, sub)
                else:
                    t = _exr_transform(raw); data = _exr_rle(t) if comp == 1 else zlib.compress(t)
                    if len(data) >= len(raw): data = raw
                chunks.append(struct.pack("<iiiii", tx, ty, l, l, len(data)) + data)
    off = len(hdr) + 8 * len(chunks); table = b""
    for c in chunks: table += struct.pack("<Q", off); off += len(c)
    open(path, "wb").write(hdr + table + b"".join(chunks))


@pytest.mark.parametrize("comp", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("tile,level_mode", [((16, 16), 0), ((7, 5), 0), ((64, 64), 0), ((8, 12), 1)])
def test_tiled_exr(tmp_path, comp, tile, level_mode):
    """Single-part tiled OpenEXR files: every codec per tile, tiles that do not divide the image (narrower / shorter edge tiles), a tile larger than the image, half and float
    channels with an extra channel between them, a data window that does not start at the origin, and a mip-mapped file of which only level 0 is the image."""
    rng = np.random.default_rng(100 * comp + tile[0])
    h, w = 23, 37
    R = (rng.random((h, w), np.float32) * 9).astype(np.float16); G = (rng.random((h, w), np.float32) * 1e4).astype(np.float32); B = (rng.random((h, w), np.float32)).astype(np.float16)
    A = np.full((h, w), 0.5, np.float16)
    write_exr_tiled(tmp_path / "t.exr", {"R": R, "G": G, "B": B, "A": A}, comp, tile, level_mode, origin=(3, -2))
    img = pt.read_float_image(tmp_path / "t.exr")
    assert img.shape == (h, w, 3)
    assert np.array_equal(img[..., 0], R.astype(np.float32)) and np.array_equal(img[..., 1], G) and np.array_equal(img[..., 2], B.astype(np.float32))


def test_tiled_exr_damage(tmp_path):
    rng = np.random.default_rng(4); img = (rng.random((20, 24), np.float32) * 5).astype(np.float16)
    write_exr_tiled(tmp_path / "t.exr", {"R": img, "G": img, "B": img}, 3, (8, 8)); src = open(tmp_path / "t.exr", "rb").read()
    ok = 0
    for k in range(300):
        d = bytearray(src); n = int(rng.integers(1, 4))
        for _ in range(n): d[int(rng.integers(8, len(d)))] = int(rng.integers(0, 256))
        if k % 5 == 0: d = d[:int(rng.integers(8, len(d)))]
        open(tmp_path / "d.exr", "wb").write(bytes(d))
        try:
            out = pt.read_float_image(tmp_path / "d.exr"); ok += 1; assert out.shape[2] == 3
        except pt.PtError as e:
            assert e.code in (4, 5)
    assert ok < 300


def write_hdr(path, rgbe, rle):
    h, w = rgbe.shape[:2]
    out = bytearray(b"#?RADIANCE\n# written by tests/test_hdr_images.py\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n" % (h, w))
    for y in range(h):
        if not rle: out += rgbe[y].tobytes(); continue
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            row = rgbe[y, :, c]; x = 0
            while x < w:
                r = 1
                while x + r < w and row[x + r] == row[x] and r < 127: r += 1
                if r >= 4: out += bytes([128 + r, row[x]]); x += r
                else:
                    k = x
                    while k < w and k - x < 128 and not (k + 3 < w and row[k] == row[k + 1] == row[k + 2] == row[k + 3]): k += 1
                    out += bytes([k - x]) + row[x:k].tobytes(); x = k
    open(path, "wb").write(bytes(out))


def test_third_party_exr_equals_a_numpy_decode_of_its_bytes():
    d = open(GOLD, "rb").read()
    a = pt.read_float_image(GOLD)
    assert a.shape == (16, 16, 3) and 0.0 <= a.min() and a.max() == 1.0
    # its header (parsed by hand once: 331 bytes, channels A B G R half, uncompressed, 16 lines) -> offset table of 16 x 8 bytes -> per line: y, size, A | B | G | R rows
    p = 331 + 16 * 8; want = np.zeros((16, 16, 3), np.float32)
    for y in range(16):
        yy, size = struct.unpack_from("<ii", d, p); assert size == 16 * 2 * 4; p += 8
        row = np.frombuffer(d, np.float16, 64, p).astype(np.float32).reshape(4, 16); p += size
        want[yy, :, 0], want[yy, :, 1], want[yy, :, 2] = row[3], row[2], row[1]
    assert np.array_equal(a, want)


@pytest.mark.parametrize("comp", [0, 1, 2, 3])
@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_exr_round_trip(tmp_path, comp, dtype):
    rng = np.random.default_rng(comp * 7 + (1 if dtype == np.float16 else 2))
    h, w = 37, 53                                           # not a multiple of the 16-line ZIP block
    img = (rng.random((h, w, 3), np.float32) ** 3 * 200.0).astype(dtype)
    img[5:9, 10:40] = dtype(1.5)                            # runs, so that the RLE and ZIP blocks really are smaller than the raw ones
    chans = {"R": img[..., 0], "G": img[..., 1], "B": img[..., 2], "A": np.ones((h, w), dtype), "Z": rng.random((h, w), np.float32)}
    path = tmp_path / "t.exr"; write_exr(path, chans, comp, origin=(-3, 11))
    got = pt.read_float_image(path)
    assert got.shape == (h, w, 3) and np.array_equal(got, img.astype(np.float32))


@pytest.mark.parametrize("case", ["half_runs", "float_noise", "half_16bit_wavelet", "tiny", "one_line_blocks"])
def test_exr_piz_round_trip(tmp_path, case):
    """PIZ (compression 4, 32-line blocks): the decoder in pt_hdrimage.cpp against the independent encoder above — flat regions (the run-length symbol), fp32 channels (two 16-bit
    planes per pixel), a block with more than 2^14 distinct values (the 16-bit wavelet), images smaller than the wavelet's first level, a height that leaves a one-line last block."""
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    if case == "half_runs":
        h, w = 70, 45; img = (rng.random((h, w, 3), np.float32) ** 2 * 30.0).astype(np.float16); img[10:50, 5:40] = np.float16(0.75); extra = {"A": np.ones((h, w), np.float16)}
    elif case == "float_noise":
        h, w = 40, 33; img = (rng.random((h, w, 3), np.float32) * 1e3).astype(np.float32); extra = {"Z": rng.random((h, w), np.float32)}
    elif case == "half_16bit_wavelet":
        h, w = 33, 700; img = rng.integers(0, 0x7BFF, (h, w, 3)).astype(np.uint16).view(np.float16); extra = {}
    elif case == "tiny":
        h, w = 3, 2; img = (rng.random((h, w, 3), np.float32)).astype(np.float16); extra = {}
    else:
        h, w = 65, 17; img = (rng.random((h, w, 3), np.float32) * 4.0).astype(np.float16); img[:, 3:9] = np.float16(2.0); extra = {}
    chans = {"R": img[..., 0], "G": img[..., 1], "B": img[..., 2]}; chans.update(extra)
    path = tmp_path / "p.exr"; write_exr(path, chans, 4, origin=(2, -5))
    raw_size = sum(c.nbytes for c in chans.values())
    if case == "half_runs": assert os.path.getsize(path) < raw_size      # the blocks really are PIZ-coded, not stored
    got = pt.read_float_image(path)
    assert got.shape == (h, w, 3) and np.array_equal(got.view(np.uint32), img.astype(np.float32).view(np.uint32))


def test_exr_piz_damaged_blocks_fail_cleanly(tmp_path):
    rng = np.random.default_rng(5); h, w = 40, 30
    img = (rng.random((h, w, 3), np.float32) * 8.0).astype(np.float16); img[:, :10] = np.float16(1.0)
    path = tmp_path / "p.exr"; write_exr(path, {"R": img[..., 0], "G": img[..., 1], "B": img[..., 2]}, 4)
    d = bytearray(open(path, "rb").read())
    for k in range(40):      # flip bytes inside the coded data: an error code or a (wrong) picture, never a crash
        e = bytearray(d); e[len(e) - 1 - int(rng.integers(0, len(e) // 2))] ^= int(rng.integers(1, 256)); q = tmp_path / ("d%d.exr" % k); open(q, "wb").write(bytes(e))
        try: pt.read_float_image(q)
        except pt.PtError: pass


def test_exr_luminance_only_and_long_names(tmp_path):
    y = np.linspace(0, 4, 20 * 8, dtype=np.float32).reshape(8, 20).astype(np.float16)
    write_exr(tmp_path / "y.exr", {"Y": y}, 2, version=2 | 0x400)
    got = pt.read_float_image(tmp_path / "y.exr")
    assert np.array_equal(got[..., 0], y.astype(np.float32)) and np.array_equal(got[..., 0], got[..., 2])


@pytest.mark.parametrize("rle", [False, True])
def test_radiance_hdr(tmp_path, rle):
    rng = np.random.default_rng(3)
    h, w = 12, 40
    rgbe = rng.integers(0, 256, (h, w, 4)).astype(np.uint8); rgbe[..., 3] = rng.integers(120, 140, (h, w)); rgbe[2, 5:30] = (10, 20, 30, 129); rgbe[4, 0] = (9, 9, 9, 0)
    write_hdr(tmp_path / "t.hdr", rgbe, rle)
    got = pt.read_float_image(tmp_path / "t.hdr")
    want = rgbe[..., :3].astype(np.float32) * np.exp2(rgbe[..., 3:4].astype(np.float32) - 136.0); want[rgbe[..., 3] == 0] = 0
    assert np.array_equal(got, want.astype(np.float32))


def test_refused_files(tmp_path):
    img = np.ones((4, 4), np.float16)
    def code(path):
        with pytest.raises(pt.PtError) as e: pt.read_float_image(path)
        return e.value.code
    write_exr(tmp_path / "pxr24.exr", {"R": img, "G": img, "B": img}, 0); d = bytearray(open(tmp_path / "pxr24.exr", "rb").read())
    i = d.index(b"compression\0compression\0") + 24 + 4; d[i] = 5; open(tmp_path / "pxr24.exr", "wb").write(d)      # PXR24 (B44, DWA likewise): lossy codecs nobody keeps radiance in
    assert code(tmp_path / "pxr24.exr") == 5                # PT_ERROR_UNSUPPORTED
    write_exr(tmp_path / "tiled.exr", {"R": img, "G": img, "B": img}, 0, version=2 | 0x200); assert code(tmp_path / "tiled.exr") == 4      # says "tiled" but has no tile description: malformed
    write_exr(tmp_path / "deep.exr", {"R": img, "G": img, "B": img}, 0, version=2 | 0x800); assert code(tmp_path / "deep.exr") == 5
    write_exr(tmp_path / "multi.exr", {"R": img, "G": img, "B": img}, 0, version=2 | 0x1000); assert code(tmp_path / "multi.exr") == 5
    write_exr(tmp_path / "nocolour.exr", {"Z": img.astype(np.float32)}, 0); assert code(tmp_path / "nocolour.exr") == 5
    open(tmp_path / "junk.bin", "wb").write(b"not an image at all"); assert code(tmp_path / "junk.bin") == 5
    assert code(tmp_path / "missing.exr") == 4              # PT_ERROR_IO
    open(tmp_path / "empty.exr", "wb").write(b""); assert code(tmp_path / "empty.exr") in (4, 5)


def test_damaged_files_fail_cleanly(tmp_path):
    rng = np.random.default_rng(11)
    img = (rng.random((20, 24, 3), np.float32) * 5).astype(np.float16)
    srcs = []
    for comp in (0, 1, 3):
        p = tmp_path / ("s%d.exr" % comp); write_exr(p, {"R": img[..., 0], "G": img[..., 1], "B": img[..., 2]}, comp); srcs.append(open(p, "rb").read())
    rgbe = rng.integers(0, 256, (6, 16, 4)).astype(np.uint8); write_hdr(tmp_path / "s.hdr", rgbe, True); srcs.append(open(tmp_path / "s.hdr", "rb").read())
    ok = 0
    for k in range(600):
        d = bytearray(srcs[k % len(srcs)])
        kind = k % 3
        if kind == 0: d = d[:int(rng.integers(0, len(d)))]                                             # truncated
        elif kind == 1:
            for _ in range(int(rng.integers(1, 6))): d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))      # flipped bytes
        else:
            i = int(rng.integers(0, len(d) - 4)); d[i:i + 4] = struct.pack("<i", int(rng.integers(-2 ** 31, 2 ** 31 - 1)))      # a wild 32-bit field
        path = tmp_path / "d.bin"; open(path, "wb").write(d)
        try:
            a = pt.read_float_image(path); ok += 1
            assert a.ndim == 3 and a.shape[2] == 3 and a.shape[0] <= 32768 and a.shape[1] <= 32768
        except pt.PtError as e:
            assert e.code in (4, 5)
    assert ok > 0                                            # (some mutations only touch pixel data)


def test_radiance_flipped_orientations(tmp_path):
    """The resolution string's sign letters: "+Y" stores the bottom row first, "-X" right to left; the transposed forms are refused."""
    rng = np.random.default_rng(31); rgbe = rng.integers(1, 256, (5, 9, 4), dtype=np.uint8)
    write_hdr(tmp_path / "std.hdr", rgbe, False); std = pt.read_float_image(tmp_path / "std.hdr")
    raw = open(tmp_path / "std.hdr", "rb").read()
    for res, want in ((b"+Y 5 +X 9", std[::-1]), (b"-Y 5 -X 9", std[:, ::-1]), (b"+Y 5 -X 9", std[::-1, ::-1])):
        (tmp_path / "o.hdr").write_bytes(raw.replace(b"-Y 5 +X 9", res)); assert np.array_equal(pt.read_float_image(tmp_path / "o.hdr"), want)
    (tmp_path / "t.hdr").write_bytes(raw.replace(b"-Y 5 +X 9", b"+X 9 -Y 5"))
    with pytest.raises(pt.PtError): pt.read_float_image(tmp_path / "t.hdr")
