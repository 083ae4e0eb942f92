"""TEST INFRASTRUCTURE: a small FLAC *encoder* in pure Python, written from the published format description (RFC 9639), used
only by tests/test_flac_cpu.py to produce streams that exercise the decoder paths the reference's fixture (test/jfk.flac:
24-bit stereo, LPC subframes, Rice partitions) does not reach -- CONSTANT / VERBATIM / FIXED subframes of every order, Rice2,
escaped partitions, wasted bits, every stereo decorrelation, 8 / 12 / 16 / 20 / 24 / 32-bit samples, variable block sizes,
an unknown total sample count, trailing bytes.  No encoder exists in the image (no ffmpeg / soundfile / flac binary)."""
import hashlib
from typing import List, Optional, Sequence

import numpy as np


class BitWriter:
    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value: int, bits: int):
        if bits == 0:
            return
        value &= (1 << bits) - 1
        self.acc = (self.acc << bits) | value
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.buf.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def unary(self, q: int):
        while q >= 32:
            self.put(0, 32)
            q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bytes(self) -> bytes:
        assert self.n == 0
        return bytes(self.buf)


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def _utf8_number(v: int) -> bytes:
    if v < 0x80:
        return bytes([v])
    out = []
    n = 1
    while True:
        n += 1
        if v < (1 << (5 * n + 1)):          # n bytes carry 7 - n + 6 (n - 1) = 5 n + 1 bits
            break
    for i in range(n - 1):
        out.append(0x80 | ((v >> (6 * i)) & 0x3F))
    lead = ((0xFF << (8 - n)) & 0xFF) | (v >> (6 * (n - 1)))
    return bytes([lead] + out[::-1])


def _residual(bw: BitWriter, res: Sequence[int], bs: int, order: int, method: int, porder: int, escape_parts=()):
    bw.put(method, 2)
    bw.put(porder, 4)
    pbits = 5 if method else 4
    parts = 1 << porder
    i = 0
    for p in range(parts):
        cnt = (bs >> porder) - (order if p == 0 else 0) if porder else bs - order
        chunk = [int(v) for v in res[i:i + cnt]]
        i += cnt
        if p in escape_parts:
            need = max([1] + [(v if v >= 0 else ~v).bit_length() + 1 for v in chunk])
            bw.put((1 << pbits) - 1, pbits)
            bw.put(need, 5)
            for v in chunk:
                bw.put(v, need)
            continue
        zz = [(v << 1) if v >= 0 else ((-v) << 1) - 1 for v in chunk]
        mean = (sum(zz) / max(len(zz), 1)) if zz else 0
        k = max(0, min((1 << pbits) - 2, int(np.log2(mean + 1))))
        bw.put(k, pbits)
        for z in zz:
            bw.unary(z >> k)
            bw.put(z, k)


def _subframe(bw: BitWriter, x: np.ndarray, bps: int, spec: dict):
    """spec: kind = constant | verbatim | fixed (order) | lpc (coefs, precision, shift); wasted; rice method / porder / escape_parts"""
    x = [int(v) for v in x]
    bs = len(x)
    wasted = int(spec.get("wasted", 0))
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in x)
        x = [v >> wasted for v in x]
    bps -= wasted
    kind = spec["kind"]
    code = {"constant": 0, "verbatim": 1}.get(kind)
    if kind == "fixed":
        code = 8 + spec["order"]
    elif kind == "lpc":
        code = 32 + len(spec["coefs"]) - 1
    bw.put(0, 1)
    bw.put(code, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        assert all(v == x[0] for v in x)
        bw.put(x[0], bps)
    elif kind == "verbatim":
        for v in x:
            bw.put(v, bps)
    elif kind == "fixed":
        order = spec["order"]
        co = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        for v in x[:order]:
            bw.put(v, bps)
        res = [x[i] - sum(c * x[i - 1 - j] for j, c in enumerate(co)) for i in range(order, bs)]
        _residual(bw, res, bs, order, spec.get("method", 0), spec.get("porder", 0), spec.get("escape_parts", ()))
    elif kind == "lpc":
        co, prec, shift = spec["coefs"], spec["precision"], spec["shift"]
        order = len(co)
        for v in x[:order]:
            bw.put(v, bps)
        bw.put(prec - 1, 4)
        bw.put(shift, 5)
        for c in co:
            bw.put(c, prec)
        res = [x[i] - (sum(c * x[i - 1 - j] for j, c in enumerate(co)) >> shift) for i in range(order, bs)]
        _residual(bw, res, bs, order, spec.get("method", 0), spec.get("porder", 0), spec.get("escape_parts", ()))
    else:
        raise ValueError(kind)


_BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
_SZ_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def encode(pcm: np.ndarray, sr: int, bps: int, block_sizes: Sequence[int], specs: List[List[dict]], stereo_modes=None,
           total_known: bool = True, with_md5: bool = True, size_code_from_streaminfo: bool = False, trailing: bytes = b"",
           extra_metadata: Optional[bytes] = None, variable: bool = False) -> bytes:
    """pcm int [frames, channels]; block_sizes: the frames' lengths (their sum = frames); specs[f][c] = subframe spec of channel c
    in frame f; stereo_modes[f] in {None, 'ls', 'sr', 'ms'} (two channels only)."""
    pcm = np.asarray(pcm, dtype=np.int64)
    n, C = pcm.shape
    assert sum(block_sizes) == n
    width = (bps + 7) // 8
    raw = b"".join(int(v).to_bytes(width, "little", signed=True) for v in pcm.reshape(-1))
    md5 = hashlib.md5(raw).digest() if with_md5 else bytes(16)
    si = BitWriter()
    si.put(min(block_sizes), 16)
    si.put(max(block_sizes), 16)
    si.put(0, 24)
    si.put(0, 24)
    si.put(sr, 20)
    si.put(C - 1, 3)
    si.put(bps - 1, 5)
    si.put(n if total_known else 0, 36)
    body = si.bytes() + md5
    out = bytearray(b"fLaC")
    blocks = [(0, body)]
    if extra_metadata is not None:
        blocks.append((4, extra_metadata))           # a VORBIS_COMMENT-typed block the decoder must skip by length
    for i, (t, b) in enumerate(blocks):
        out.append((0x80 if i == len(blocks) - 1 else 0) | t)
        out += len(b).to_bytes(3, "big")
        out += b
    pos = 0
    for f, bs in enumerate(block_sizes):
        x = pcm[pos:pos + bs]
        mode = (stereo_modes or [None] * len(block_sizes))[f]
        bw = BitWriter()
        bw.put(0b11111111111110, 14)
        bw.put(0, 1)
        bw.put(1 if variable else 0, 1)
        code = _BS_CODES.get(bs)
        if code is None:
            code = 6 if bs <= 256 else 7
        bw.put(code, 4)
        bw.put(0, 4)                                   # sample rate: from STREAMINFO
        bw.put({None: C - 1, "ls": 8, "sr": 9, "ms": 10}[mode], 4)
        bw.put(0 if size_code_from_streaminfo else _SZ_CODES[bps], 3)
        bw.put(0, 1)
        for b in _utf8_number(pos if variable else f):
            bw.put(b, 8)
        if code == 6:
            bw.put(bs - 1, 8)
        elif code == 7:
            bw.put(bs - 1, 16)
        hdr = bw.bytes()
        bw.put(crc8(hdr), 8)
        chans = [x[:, c] for c in range(C)]
        widths = [bps] * C
        if mode == "ls":
            chans = [x[:, 0], x[:, 0] - x[:, 1]]
            widths = [bps, bps + 1]
        elif mode == "sr":
            chans = [x[:, 0] - x[:, 1], x[:, 1]]
            widths = [bps + 1, bps]
        elif mode == "ms":
            chans = [(x[:, 0] + x[:, 1]) >> 1, x[:, 0] - x[:, 1]]
            widths = [bps, bps + 1]
        for c in range(C):
            _subframe(bw, chans[c], widths[c], specs[f][c])
        bw.align()
        frame = bw.bytes()
        out += frame + crc16(frame).to_bytes(2, "big")
        pos += bs
    return bytes(out) + trailing
