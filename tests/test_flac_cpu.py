"""FLAC front-end (the reference decodes every container through ffmpeg, audio/utils.py:63-125, and its one
real-speech fixture is test/jfk.flac).  The decoder is host code of libswx.so (csrc/swx_flac.hip) -- no GPU is involved, so
these run in the CPU suite: (1) the reference's own fixture against the MD5 signature its encoder left in STREAMINFO,
(2) streams made by tests/flac_encoder.py (test infrastructure) that reach every subframe type / residual coding / stereo mode /
sample size, bit-exact against the samples that were encoded, (3) damaged streams -> RuntimeError, (4) the loader paths."""
import hashlib
import zlib
import os

import numpy as np
import pytest
import torch

from stable_ts_amd import audio_io

import flac_encoder as fe

JFK = "/root/reference/test/jfk.flac"
HERE = os.path.dirname(os.path.abspath(__file__))
JFK16 = os.path.join(HERE, "golden", "jfk_16k_mono.flac")


def _decode_int(data: bytes):
    import ctypes
    from stable_ts_amd import _lib
    lib = _lib.load()
    info = _lib.swx_flac_info()
    n = lib.swx_flac_decode(data, len(data), None, 0, ctypes.byref(info))
    if n < 0:
        return n, None, info
    out = np.empty((n, info.channels), np.int32)
    n2 = lib.swx_flac_decode(data, len(data), out.ctypes.data, n, ctypes.byref(info))
    assert n2 == n
    return n, out, info


@pytest.mark.skipif(not os.path.exists(JFK), reason="reference fixture not present (GPU box)")
def test_reference_fixture_matches_its_md5_signature():
    info = audio_io.flac_info(JFK)
    assert (info["sr"], info["channels"], info["bits"], info["frames"]) == (44100, 2, 24, 485100)
    assert any(info["md5"])
    data = open(JFK, "rb").read()
    n, pcm, ci = _decode_int(data)
    assert n == 485100
    raw = pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    assert hashlib.md5(raw).digest() == info["md5"]            # the encoder's signature of the unencoded samples
    x, sr = audio_io.read_flac(JFK)                            # checks the same signature itself
    assert sr == 44100 and x.shape == (485100, 2) and x.dtype == np.float32
    assert np.array_equal(x, (pcm.astype(np.float64) / (1 << 23)).astype(np.float32))
    assert 0.05 < float(np.abs(x).max()) <= 1.0
    # loader: 16 kHz mono on the s16 grid, 11 s
    y = audio_io.load_audio(JFK)
    assert y.dtype == np.float32 and abs(len(y) - 176000) <= 1
    assert np.array_equal(y, np.round(y * 32768.0) / 32768.0)
    assert audio_io.get_metadata(JFK) == dict(sr=44100, duration=485100 / 44100)
    # streamed source == whole-file source (AudioLoader contract)
    a = audio_io.AudioLoader(JFK, stream=True, new_chunk_divisor=None)
    b = audio_io.AudioLoader(JFK, stream=False, new_chunk_divisor=None)
    ca, cb = a.next_chunk(0, 480000), b.next_chunk(0, 480000)
    assert ca.shape == cb.shape and float((ca - cb).abs().max()) <= 1.0 / 32768.0 + 1e-9
    a.terminate()
    b.terminate()


@pytest.mark.skipif(not (os.path.exists(JFK) and os.path.exists(JFK16)), reason="fixtures not present")
def test_committed_16k_fixture_is_the_loaded_reference_fixture():
    """tests/golden/jfk_16k_mono.flac (made by tests/golden/make_jfk_fixture.py) holds exactly what load_audio() yields for the
    reference's fixture, so the GPU box -- which has no /root/reference -- feeds the path the same real speech"""
    y = audio_io.load_audio(JFK)
    z = audio_io.load_audio(JFK16)
    assert np.array_equal(y, z)


def _rng_pcm(rng, n, C, bps, smooth=True):
    amp = (1 << (bps - 1)) - 1
    if smooth:
        t = np.arange(n)[:, None]
        x = 0.4 * amp * np.sin(2 * np.pi * t * (0.003 + 0.002 * np.arange(C)[None, :])) + rng.normal(0, amp * 0.01, (n, C))
    else:
        x = rng.integers(-amp - 1, amp + 1, (n, C))
    return np.clip(np.round(x), -amp - 1, amp).astype(np.int64)


CASES = []
for bps in (8, 12, 16, 20, 24, 32):
    CASES.append(dict(name=f"fixed-orders-{bps}bit", bps=bps, C=1, blocks=[256] * 5,
                      specs=[[dict(kind="fixed", order=o, porder=2)] for o in range(5)]))
CASES += [
    dict(name="verbatim+constant", bps=16, C=2, blocks=[192, 192], const=True,
         specs=[[dict(kind="verbatim"), dict(kind="constant")], [dict(kind="constant"), dict(kind="verbatim")]]),
    dict(name="lpc-orders", bps=16, C=1, blocks=[1024] * 3,
         specs=[[dict(kind="lpc", coefs=[1900, -900], precision=12, shift=10, porder=3)],
                [dict(kind="lpc", coefs=[3000, -2900, 900, 20, -10, 5, 3, -2], precision=14, shift=10, porder=0)],
                [dict(kind="lpc", coefs=[(-1) ** j * (40 - j) for j in range(32)], precision=15, shift=9, porder=4, method=1)]]),
    dict(name="rice2+escape", bps=24, C=1, blocks=[512, 512],
         specs=[[dict(kind="fixed", order=2, method=1, porder=3, escape_parts=(1, 5))],
                [dict(kind="fixed", order=1, method=0, porder=2, escape_parts=(0, 3))]]),
    dict(name="wasted-bits", bps=16, C=2, blocks=[576], wasted=3,
         specs=[[dict(kind="fixed", order=2, wasted=3), dict(kind="verbatim", wasted=3)]]),
    dict(name="stereo-modes", bps=16, C=2, blocks=[256] * 4, modes=[None, "ls", "sr", "ms"],
         specs=[[dict(kind="fixed", order=2, porder=1), dict(kind="fixed", order=2, porder=1)]] * 4),
    dict(name="stereo-modes-24", bps=24, C=2, blocks=[4096, 1000], modes=["ms", "ls"],
         specs=[[dict(kind="fixed", order=3, porder=4), dict(kind="fixed", order=1, porder=0)],
                [dict(kind="verbatim"), dict(kind="fixed", order=2, porder=0)]]),
    dict(name="stereo-32bit-side-is-33-bits", bps=32, C=2, blocks=[192], modes=["ms"], noise=True,
         specs=[[dict(kind="verbatim"), dict(kind="verbatim")]]),
    dict(name="odd-block-sizes", bps=16, C=1, blocks=[200, 3000, 17], specs=[[dict(kind="fixed", order=2)]] * 3, variable=True),
    dict(name="unknown-total+no-md5+sizecode0", bps=16, C=1, blocks=[512, 100], specs=[[dict(kind="fixed", order=1)]] * 2,
         total_known=False, with_md5=False, size_code0=True),
    dict(name="eight-channels", bps=16, C=8, blocks=[256], specs=[[dict(kind="fixed", order=o % 5) for o in range(8)]]),
    dict(name="metadata-skipped+trailing", bps=16, C=1, blocks=[256], specs=[[dict(kind="fixed", order=2)]],
         extra=b"\x05\x00\x00\x00hello\x00\x00\x00\x00", trailing=b"TAG" + bytes(125)),
    dict(name="noise-large-residuals", bps=24, C=1, blocks=[1024], noise=True, specs=[[dict(kind="fixed", order=4, porder=2)]]),
]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_synthetic_streams_decode_bit_exact(case):
    rng = np.random.default_rng(zlib.crc32(case["name"].encode()))
    n = sum(case["blocks"])
    pcm = _rng_pcm(rng, n, case["C"], case["bps"], smooth=not case.get("noise"))
    if case.get("wasted"):
        pcm = (pcm >> case["wasted"]) << case["wasted"]
    if case.get("const"):
        pos = 0
        for f, bs in enumerate(case["blocks"]):
            for c, sp in enumerate(case["specs"][f]):
                if sp["kind"] == "constant":
                    pcm[pos:pos + bs, c] = pcm[pos, c]
            pos += bs
    data = fe.encode(pcm, 22050, case["bps"], case["blocks"], case["specs"], stereo_modes=case.get("modes"),
                     total_known=case.get("total_known", True), with_md5=case.get("with_md5", True),
                     size_code_from_streaminfo=case.get("size_code0", False), trailing=case.get("trailing", b""),
                     extra_metadata=case.get("extra"), variable=case.get("variable", False))
    n_dec, out, info = _decode_int(data)
    assert n_dec == n, n_dec
    assert (info.sample_rate, info.channels, info.bits_per_sample) == (22050, case["C"], case["bps"])
    assert np.array_equal(out.astype(np.int64), pcm)
    x, sr = audio_io.read_flac(data)                       # MD5 path + scaling
    assert sr == 22050
    assert np.array_equal(x, (pcm.astype(np.float64) / float(1 << (case["bps"] - 1))).astype(np.float32))


def _small_stream():
    rng = np.random.default_rng(3)
    pcm = _rng_pcm(rng, 1024, 2, 16)
    data = fe.encode(pcm, 16000, 16, [512, 512], [[dict(kind="fixed", order=2, porder=2)] * 2] * 2, stereo_modes=["ms", None])
    return pcm, data


def test_damaged_streams_raise():
    pcm, data = _small_stream()
    with pytest.raises(RuntimeError, match="not a FLAC stream"):
        audio_io.read_flac(b"RIFF" + data[4:])
    with pytest.raises(RuntimeError, match="truncated"):
        audio_io.read_flac(data[: len(data) - 40])
    hits = 0
    for pos in range(60, len(data) - 2, 37):               # any flipped bit inside a frame is caught by a CRC or a structural check
        bad = bytearray(data)
        bad[pos] ^= 0x10
        with pytest.raises(RuntimeError):
            audio_io.read_flac(bytes(bad))
        hits += 1
    assert hits > 20
    # a stream whose samples do not match the MD5 signature (signature damaged, frames intact)
    bad = bytearray(data)
    bad[4 + 4 + 18] ^= 0xFF
    with pytest.raises(RuntimeError, match="MD5"):
        audio_io.read_flac(bytes(bad))
    assert audio_io.read_flac(bytes(bad), verify_md5=False)[0].shape == (1024, 2)


def test_loader_paths_take_flac(tmp_path):
    pcm, data = _small_stream()
    # 16 kHz stereo s16 -> the loader's mono mix on the s16 grid, no resampling involved
    want = audio_io.to_s16((pcm.astype(np.float64) / 32768.0).mean(axis=1).astype(np.float32)).astype(np.float32) / 32768.0
    got = audio_io.load_audio(data)
    assert np.array_equal(got, want)
    p = tmp_path / "a.flac"
    p.write_bytes(data)
    assert np.array_equal(audio_io.load_audio(str(p)), want)
    assert audio_io.is_flac(str(p)) and audio_io.is_flac(data) and not audio_io.is_wav(data)
    assert audio_io.get_metadata(str(p)) == dict(sr=16000, duration=1024 / 16000)
    t = audio_io.prep_audio(str(p))
    assert torch.is_tensor(t) and np.array_equal(t.numpy(), want)
    ld = audio_io.AudioLoader(str(p), stream=True, new_chunk_divisor=None)
    ch = ld.next_chunk(0, 1024)
    assert np.array_equal(ch.numpy(), want)
    ld.terminate()
    # ID3v2 tag in front of the marker
    tagged = b"ID3\x04\x00\x00" + bytes([0, 0, 0, 20]) + bytes(20) + data
    assert audio_io.is_flac(tagged) and np.array_equal(audio_io.load_audio(tagged), want)


def test_forged_streaminfo_frame_count_is_refused_before_any_allocation():
    """STREAMINFO's 36-bit sample count is untrusted input -- 2**36 - 1 frames declared by a ~1 KB stream must not
    reach np.empty (a 512 GiB request) but raise the loader's RuntimeError (audio/__init__.py:215-221: RuntimeError for
    audio that cannot be decoded)."""
    _, data = _small_stream()
    bad = bytearray(data)
    si = 8                                   # "fLaC" + the 4-byte metadata block header; STREAMINFO: total samples = low nibble of
    bad[si + 13] |= 0x0F                     # byte 13 + bytes 14..17
    bad[si + 14: si + 18] = b"\xff\xff\xff\xff"
    with pytest.raises(RuntimeError, match="declares 68719476735 sample frames"):
        audio_io.read_flac(bytes(bad), verify_md5=False)
    # a count that passes the size bound but is wrong is caught after decoding
    bad = bytearray(data)
    bad[si + 17] ^= 0x01
    with pytest.raises(RuntimeError, match="sample frames it declares|Failed to load audio"):
        audio_io.read_flac(bytes(bad), verify_md5=False)


def test_is_flac_behind_an_id3_tag_never_reads_the_file(tmp_path, monkeypatch):
    """most MP3s start with an ID3v2 tag; is_flac() must decide from 10 + 4 bytes, not slurp the file"""
    p = tmp_path / "song.mp3"
    p.write_bytes(b"ID3\x04\x00\x00" + bytes([0, 0, 0x10, 0]) + bytes(2048) + b"\xff\xfb\x90\x00" + bytes(1 << 20))
    reads = []
    real_open = audio_io._open_binary

    class Spy:
        def __init__(self, f):
            self.f = f

        def __enter__(self):
            return self

        def __exit__(self, *a):
            self.f.close()

        def read(self, n=-1):
            out = self.f.read(n)
            reads.append((n, len(out)))
            return out

        def seek(self, *a):
            return self.f.seek(*a)

    monkeypatch.setattr(audio_io, "_open_binary", lambda src: Spy(real_open(src)))
    assert audio_io.is_flac(str(p)) is False
    assert reads and all(n >= 0 for n, _ in reads) and sum(got for _, got in reads) <= 14, reads
    reads.clear()
    _, data = _small_stream()
    q = tmp_path / "tagged.flac"
    q.write_bytes(b"ID3\x04\x00\x00" + bytes([0, 0, 0, 20]) + bytes(20) + data)
    assert audio_io.is_flac(str(q)) is True
    assert sum(got for _, got in reads) <= 14, reads
