"""A small FLAC ENCODER for the tests of csrc/flac_host.hip (test infrastructure only; nothing under sylber_amd/ imports it).

No FLAC file, codec library or command-line encoder exists in the build image, so the decoder's tests make their own streams.  This file is written
independently of the decoder, from the published format (xiph.org "FLAC format" / RFC 9639), and deliberately through different mechanics: table-driven CRCs,
hashlib's MD5, Python's arbitrary-precision integers for the bit stream, numpy least squares for the LPC coefficients.  It can emit every construct the decoder
claims: CONSTANT / VERBATIM / FIXED (orders 0-4) / LPC (orders 1-32, any coefficient precision and shift) subframes, wasted bits, residual coding methods 0 / 1
with any partition order and escape partitions, the four channel assignments, 4-32 bits per sample, explicit or coded block sizes and sample rates, fixed and
variable blocking, extra metadata blocks, an ID3v2 prefix, unknown length, absent MD5.  `choose(...)` callbacks let a test force each of them."""
import hashlib
import struct

import numpy as np

_CRC8 = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = ((_c << 1) ^ 0x07) & 0xff if _c & 0x80 else (_c << 1) & 0xff
    _CRC8.append(_c)
_CRC16 = []
for _i in range(256):
    _c = _i << 8
    for _ in range(8):
        _c = ((_c << 1) ^ 0x8005) & 0xffff if _c & 0x8000 else (_c << 1) & 0xffff
    _CRC16.append(_c)


def crc8(b):
    c = 0
    for x in b:
        c = _CRC8[c ^ x]
    return c


def crc16(b):
    c = 0
    for x in b:
        c = ((c << 8) & 0xffff) ^ _CRC16[(c >> 8) ^ x]
    return c


class Bits:
    """MSB-first bit string as one big integer"""
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, nbits):
        if nbits:
            self.v = (self.v << nbits) | (int(value) & ((1 << nbits) - 1))
            self.n += nbits

    def signed(self, value, nbits):
        self.put(int(value) & ((1 << nbits) - 1), nbits)

    def unary(self, zeros):
        self.put(1, zeros + 1)

    def pad(self):
        self.put(0, (-self.n) % 8)

    def bytes(self):
        assert self.n % 8 == 0
        return self.v.to_bytes(self.n // 8, "big") if self.n else b""


def _utf8(n):
    if n < 0x80:
        return bytes([n])
    out, k = [], 0
    while n >= (0x40 >> k):
        out.append(0x80 | (n & 0x3f)); n >>= 6; k += 1
    lead = ((0xff << (7 - k)) & 0xff) | n
    return bytes([lead] + out[::-1])


def rice_bits(res, k):
    u = np.where(res >= 0, 2 * res.astype(np.int64), -2 * res.astype(np.int64) - 1)
    return int((u >> k).sum()) + (k + 1) * len(res)


def put_residual(bw, res, order, bs, porder, method, escape_partition=None):
    """res: residuals of samples order..bs-1"""
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    bw.put(method, 2)
    bw.put(porder, 4)
    parts = 1 << porder
    assert bs % parts == 0 and (bs >> porder) >= order
    i = 0
    for pt in range(parts):
        cnt = (bs >> porder) - (order if pt == 0 else 0)
        r = res[i:i + cnt]; i += cnt
        if escape_partition is not None and pt == escape_partition:
            nb = 0 if cnt == 0 or not r.any() else int(max(int(r.max()).bit_length(), int(-r.min() - 1).bit_length() if r.min() < 0 else 0)) + 1
            bw.put(esc, pbits); bw.put(nb, 5)
            for x in r:
                bw.signed(int(x), nb)
            continue
        best = min(range(esc), key=lambda k: rice_bits(r, k)) if cnt else 0
        bw.put(best, pbits)
        for x in r:
            x = int(x)
            u = 2 * x if x >= 0 else -2 * x - 1
            bw.unary(u >> best)
            bw.put(u & ((1 << best) - 1), best)
    assert i == len(res)


def put_subframe(bw, s, bps, spec):
    """s: int64 samples of one channel of one block; spec = dict(kind=..., order=..., precision=..., porder=..., method=..., escape=..., wasted=...)"""
    bs = len(s)
    wasted = int(spec.get("wasted", 0))
    if wasted:
        assert not (s & ((1 << wasted) - 1)).any()
        s = s >> wasted
    kind = spec["kind"]
    order = int(spec.get("order", 0))
    bw.put(0, 1)
    bw.put({"constant": 0, "verbatim": 1, "fixed": 8 + order, "lpc": 31 + order}[kind], 6)
    if wasted:
        bw.put(1, 1); bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    b = bps - wasted
    if kind == "constant":
        assert (s == s[0]).all()
        bw.signed(int(s[0]), b)
        return
    if kind == "verbatim":
        for x in s:
            bw.signed(int(x), b)
        return
    for x in s[:order]:
        bw.signed(int(x), b)
    if kind == "fixed":
        r = s.astype(object)
        for _ in range(order):
            r = r[1:] - r[:-1]
        res = np.array(list(r), dtype=np.int64)
    else:
        prec, sv = int(spec.get("precision", 12)), s.astype(np.float64)
        A = np.stack([sv[order - 1 - j:bs - 1 - j] for j in range(order)], 1)
        coef = np.linalg.lstsq(A, sv[order:], rcond=None)[0] if bs > order else np.zeros(order)
        cmax = max(float(np.abs(coef).max()), 1e-9)
        shift = int(spec.get("shift", max(0, min(15, prec - 1 - int(np.ceil(np.log2(cmax + 1e-12)))))))
        q = np.clip(np.round(coef * (1 << shift)), -(1 << (prec - 1)), (1 << (prec - 1)) - 1).astype(np.int64)
        bw.put(prec - 1, 4)
        bw.signed(shift, 5)
        for c in q:
            bw.signed(int(c), prec)
        so = [int(x) for x in s]
        res = np.array([so[i] - (sum(int(q[j]) * so[i - 1 - j] for j in range(order)) >> shift) for i in range(order, bs)], dtype=np.int64)
    assert res.size == 0 or (int(res.min()) >= -(1 << 31) and int(res.max()) < (1 << 31))
    put_residual(bw, res, order, bs, int(spec.get("porder", 0)), int(spec.get("method", 0)), spec.get("escape"))


BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
SR_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
SS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def encode(pcm, sr, bps, blocksize=4096, choose=None, assignment=None, variable=False, explicit=False, known_length=True, with_md5=True,
           extra_blocks=(), id3=0):
    """pcm [n, channels] integers in the bps-bit signed range -> FLAC bytes.  choose(frame, channel, samples) -> subframe spec (default: fixed order 2, partition
    order 2); assignment(frame) -> 0 (independent), 8 (left/side), 9 (side/right), 10 (mid/side) for two channels; explicit: block size and sample rate coded as
    explicit fields, the sample size taken from STREAMINFO (code 0)"""
    pcm = np.asarray(pcm, dtype=np.int64)
    n, nch = pcm.shape
    frames = []
    f, pos = 0, 0
    sizes = []
    while pos < n:
        bs = min(blocksize, n - pos)
        blk = pcm[pos:pos + bs]
        code = int(assignment(f)) if (assignment and nch == 2) else nch - 1
        hb = Bits()
        hb.put(0x3ffe, 14); hb.put(0, 1); hb.put(1 if variable else 0, 1)
        bsc = BS_CODES.get(bs) if not explicit else None
        if bsc is None:
            bsc = 6 if bs <= 256 else 7
        src = SR_CODES.get(sr, 0) if not explicit else (12 if sr % 1000 == 0 and sr < 256000 else (13 if sr < 65536 else 14))
        hb.put(bsc, 4); hb.put(src, 4); hb.put(code, 4); hb.put(0 if explicit or bps not in SS_CODES else SS_CODES[bps], 3); hb.put(0, 1)
        head = hb.bytes() + _utf8(pos if variable else f)
        if bsc == 6:
            head += bytes([bs - 1])
        elif bsc == 7:
            head += struct.pack(">H", bs - 1)
        if src == 12:
            head += bytes([sr // 1000])
        elif src == 13:
            head += struct.pack(">H", sr)
        elif src == 14:
            head += struct.pack(">H", sr // 10)
        head += bytes([crc8(head)])
        if code == 8:
            chans, bits = [blk[:, 0], blk[:, 0] - blk[:, 1]], [bps, bps + 1]
        elif code == 9:
            chans, bits = [blk[:, 0] - blk[:, 1], blk[:, 1]], [bps + 1, bps]
        elif code == 10:
            chans, bits = [(blk[:, 0] + blk[:, 1]) >> 1, blk[:, 0] - blk[:, 1]], [bps, bps + 1]
        else:
            chans, bits = [blk[:, c] for c in range(nch)], [bps] * nch
        bw = Bits()
        for c, (s, b) in enumerate(zip(chans, bits)):
            spec = choose(f, c, s) if choose else {"kind": "fixed", "order": min(2, bs), "porder": 2 if bs % 4 == 0 and bs // 4 >= 2 else 0}
            put_subframe(bw, s, b, spec)
        bw.pad()
        body = head + bw.bytes()
        frames.append(body + struct.pack(">H", crc16(body)))
        sizes.append(len(frames[-1]))
        pos += bs; f += 1
    w = (bps + 7) // 8
    raw = b"".join(int(x).to_bytes(w, "little", signed=True) for x in pcm.reshape(-1))
    md5 = hashlib.md5(raw).digest() if with_md5 else bytes(16)
    si = struct.pack(">HH", min(blocksize, 65535), min(blocksize, 65535)) + min(sizes).to_bytes(3, "big") + max(sizes).to_bytes(3, "big")
    packed = (sr << 44) | ((nch - 1) << 41) | ((bps - 1) << 36) | (n if known_length else 0)
    si += packed.to_bytes(8, "big") + md5
    blocks = [(0, si)] + list(extra_blocks)
    out = b""
    if id3:
        out += b"ID3\x04\x00\x00" + bytes([(id3 >> 21) & 0x7f, (id3 >> 14) & 0x7f, (id3 >> 7) & 0x7f, id3 & 0x7f]) + bytes(id3)
    out += b"fLaC"
    for i, (t, payload) in enumerate(blocks):
        out += bytes([(0x80 if i == len(blocks) - 1 else 0) | t]) + len(payload).to_bytes(3, "big") + payload
    return out + b"".join(frames)
