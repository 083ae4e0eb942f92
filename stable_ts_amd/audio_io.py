"""Audio front-end: file decoding, resampling, the voice-band filter and ``AudioLoader``.

Mirrors the part of ``stable_whisper/audio`` the window loop sits on:

* ``load_audio`` (audio/utils.py:63-125): any source -> mono f32 waveform at 16 kHz on the s16 grid.  The reference pipes
  everything through an ``ffmpeg -f s16le -ac 1 -ar 16000`` child process; offline boxes have no ffmpeg, so RIFF/WAVE
  files (PCM 8/16/24/32 bit, IEEE float, WAVE_FORMAT_EXTENSIBLE) and native FLAC streams (``read_flac``: the host-side
  decoder of libswx.so, ``csrc/swx_flac.hip``, with the encoder's MD5 signature checked -- the reference's real-speech
  fixture is ``test/jfk.flac``) are decoded here and other containers go to ffmpeg with the reference's command line
  when it is on PATH.  A 16 kHz mono s16 WAV yields the same samples either way.
* ``resample`` (audio/utils.py:128-129 -> ``torchaudio.functional.resample`` defaults): Hann-windowed sinc
  interpolation, 6 zero crossings, roll-off 0.99, evaluated as one strided convolution.  torchaudio is not installed
  here, so its published algorithm is restated; ``resample_blocks`` is the same filter run block-wise for streaming.
* ``voice_freq_filter`` (audio/utils.py:132-149): 5 kHz low-pass then 200 Hz high-pass biquads (Q = 0.707), output
  clamped to [-1, 1] after each stage like ``torchaudio.functional.lfilter(clamp=True)``.
* ``prep_audio`` (audio/__init__.py:74-149) and ``AudioLoader`` (audio/__init__.py:152-638): the buffered
  seek / chunk / section contract ``transcribe`` drives (``next_chunk``, ``next_valid_chunk``, ``skip_to_next_section``,
  ``negate_ts_sections``, duration and sample-count estimates, ``new_chunk_divisor`` rounding, s16le -> f32 conversion,
  post-prep callback), for in-memory sources and for streamed sources.  ``tests/test_audio_cpu.py`` drives this class
  and the reference's with the same call sequences and compares every returned chunk.

Out of scope: denoisers (demucs / dfnet / noisereduce need their models), yt-dlp URLs.
"""
import io
import math
import shutil
import struct
import subprocess
import warnings
import wave
from typing import BinaryIO, Callable, Iterator, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from .audio import SAMPLE_RATE

# ------------------------------------------------------------------------------------------------- RIFF/WAVE decoding
_FMT_PCM, _FMT_FLOAT, _FMT_EXTENSIBLE = 0x0001, 0x0003, 0xFFFE


class _WavInfo:
    __slots__ = ("fmt", "channels", "sr", "block", "bits", "data_offset", "n_frames")


def _open_binary(source: Union[str, bytes]) -> BinaryIO:
    return io.BytesIO(source) if isinstance(source, (bytes, bytearray)) else open(source, "rb")


def is_wav(source: Union[str, bytes]) -> bool:
    try:
        with _open_binary(source) as f:
            head = f.read(12)
    except OSError:
        return False
    return len(head) == 12 and head[:4] == b"RIFF" and head[8:12] == b"WAVE"


def _parse_wav(f: BinaryIO) -> _WavInfo:
    head = f.read(12)
    if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        raise RuntimeError("Failed to load audio: not a RIFF/WAVE file")
    info = None
    while True:
        ck = f.read(8)
        if len(ck) < 8:
            raise RuntimeError("Failed to load audio: WAVE file without a data chunk")
        tag, size = ck[:4], struct.unpack("<I", ck[4:])[0]
        if tag == b"fmt ":
            body = f.read(size + (size & 1))
            fmt, ch, sr, _, block, bits = struct.unpack("<HHIIHH", body[:16])
            if fmt == _FMT_EXTENSIBLE and size >= 26:
                fmt = struct.unpack("<H", body[24:26])[0]       # first two bytes of the sub-format GUID
            ok = (fmt == _FMT_PCM and bits in (8, 16, 24, 32)) or (fmt == _FMT_FLOAT and bits in (32, 64))
            if not ok or ch < 1 or sr < 1:
                raise RuntimeError(f"Failed to load audio: unsupported WAVE format (tag 0x{fmt:04x}, {bits} bits, "
                                   f"{ch} channels, {sr} Hz)")
            info = _WavInfo()
            info.fmt, info.channels, info.sr, info.block, info.bits = fmt, ch, sr, ch * bits // 8, bits
        elif tag == b"data":
            if info is None:
                raise RuntimeError("Failed to load audio: WAVE data chunk before the fmt chunk")
            info.data_offset = f.tell()
            here = f.tell()
            f.seek(0, 2)
            avail = f.tell() - here
            f.seek(here)
            if size == 0xFFFFFFFF or size == 0 or size > avail:   # streamed / truncated writers leave it open
                size = avail
            info.n_frames = size // max(info.block, 1)
            return info
        else:
            f.seek(size + (size & 1), 1)


def _decode_frames(raw: bytes, info: _WavInfo) -> np.ndarray:
    """interleaved sample bytes -> f32 [frames, channels] in [-1, 1)"""
    fmt, bits = info.fmt, info.bits
    if fmt == _FMT_PCM:
        if bits == 8:
            x = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
        elif bits == 32:
            x = (np.frombuffer(raw, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise RuntimeError(f"Failed to load audio: unsupported PCM sample width ({bits} bits)")
    elif fmt == _FMT_FLOAT and bits in (32, 64):
        x = np.frombuffer(raw, "<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise RuntimeError(f"Failed to load audio: unsupported WAVE format tag 0x{fmt:04x} ({bits} bits)")
    return x.reshape(-1, info.channels)


def read_wav(source: Union[str, bytes]) -> Tuple[np.ndarray, int]:
    """RIFF/WAVE file or bytes -> (f32 [frames, channels], sample rate)."""
    with _open_binary(source) as f:
        info = _parse_wav(f)
        raw = f.read(info.n_frames * info.block)
    return _decode_frames(raw[: len(raw) // info.block * info.block], info), info.sr


def write_wav(path: str, audio: Union[np.ndarray, torch.Tensor], sr: int):
    """mono / [channels, n] waveform -> 16-bit PCM WAVE (``AudioLoader.save_final_audio``)."""
    a = audio.detach().cpu().numpy() if torch.is_tensor(audio) else np.asarray(audio)
    a = a[None] if a.ndim == 1 else a
    with wave.open(path, "wb") as w:
        w.setnchannels(a.shape[0])
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(to_s16(a.T.reshape(-1)).tobytes())


# ------------------------------------------------------------------------------------------------------ FLAC decoding
def _source_bytes(source: Union[str, bytes]) -> bytes:
    if isinstance(source, (bytes, bytearray)):
        return bytes(source)
    with open(source, "rb") as f:
        return f.read()


def is_flac(source: Union[str, bytes]) -> bool:
    """native FLAC stream (an ID3v2 tag in front of the marker, which some taggers write, is skipped by ``read_flac``).
    Reads 10 bytes, and 4 more behind an ID3v2 tag: never the whole file (most MP3s start with such a tag)."""
    try:
        with _open_binary(source) as f:
            head = f.read(10)
            if head[:4] == b"fLaC":
                return True
            if head[:3] != b"ID3" or len(head) < 10:
                return False
            f.seek(10 + _id3_size(head))
            return f.read(4) == b"fLaC"
    except (OSError, ValueError):
        return False


def _id3_size(head: bytes) -> int:
    """bytes of an ID3v2 tag behind its 10-byte header (sync-safe size + the optional footer)"""
    size = ((head[6] & 0x7F) << 21) | ((head[7] & 0x7F) << 14) | ((head[8] & 0x7F) << 7) | (head[9] & 0x7F)
    return size + (10 if head[5] & 0x10 else 0)


def _skip_id3(data: bytes) -> bytes:
    if data[:3] == b"ID3" and len(data) >= 10:
        return data[10 + _id3_size(data):]
    return data


def flac_info(source: Union[str, bytes]) -> dict:
    """STREAMINFO of a FLAC file / bytes: dict(sr, channels, bits, frames, md5)."""
    import ctypes
    from . import _lib
    data = _skip_id3(_source_bytes(source))
    info = _lib.swx_flac_info()
    rc = _lib.load().swx_flac_probe(data, len(data), ctypes.byref(info))
    if rc < 0:
        raise RuntimeError(f"Failed to load audio: {_lib.load().swx_strerror(rc).decode()}")
    return dict(sr=info.sample_rate, channels=info.channels, bits=info.bits_per_sample, frames=int(info.total_samples),
                md5=bytes(info.md5))


def read_flac(source: Union[str, bytes], verify_md5: bool = True) -> Tuple[np.ndarray, int]:
    """FLAC file or bytes -> (f32 [frames, channels] in [-1, 1), sample rate).  Decoded by ``swx_flac_decode`` (host code of
    libswx.so: every frame's CRC-8 / CRC-16 is verified there); the MD5 signature of the unencoded samples that the
    encoder left in STREAMINFO is checked here (all-zero = not set).  ``RuntimeError`` like the reference's loader on a
    stream that cannot be decoded (audio/utils.py:109-121)."""
    import ctypes
    import hashlib
    from . import _lib
    lib = _lib.load()
    data = _skip_id3(_source_bytes(source))
    info = _lib.swx_flac_info()
    rc = lib.swx_flac_probe(data, len(data), ctypes.byref(info))
    if rc < 0:
        raise RuntimeError(f"Failed to load audio: {lib.swx_strerror(rc).decode()}")
    frames = int(info.total_samples)
    if frames == 0:                               # a streamed encoder did not go back to fill the count in: count first
        frames = int(lib.swx_flac_decode(data, len(data), None, 0, ctypes.byref(info)))
        if frames < 0:
            raise RuntimeError(f"Failed to load audio: {lib.swx_strerror(frames).decode()}")
        total_unknown = True
    else:
        total_unknown = False
    # STREAMINFO is untrusted: a frame holds at most 65 535 samples per channel and takes at least 11 bytes (sync + header +
    # CRCs + one constant subframe per channel), so a stream of len(data) bytes cannot hold more than this many sample frames
    if frames > (len(data) // 11 + 1) * 65535:
        raise RuntimeError(f"Failed to load audio: FLAC header declares {frames} sample frames in a {len(data)}-byte stream")
    try:
        pcm = np.empty((frames, info.channels), dtype=np.int32)
    except MemoryError as e:
        raise RuntimeError(f"Failed to load audio: {frames} sample frames do not fit in memory") from e
    n = int(lib.swx_flac_decode(data, len(data), pcm.ctypes.data, frames, ctypes.byref(info)))
    if n < 0:
        raise RuntimeError(f"Failed to load audio: {lib.swx_strerror(n).decode()}")
    if n != frames and not total_unknown:
        raise RuntimeError(f"Failed to load audio: FLAC stream holds {n} of the {frames} sample frames it declares")
    pcm = pcm[:n]
    md5 = bytes(info.md5)
    if verify_md5 and any(md5):
        width = (info.bits_per_sample + 7) // 8
        raw = pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width].tobytes()
        if hashlib.md5(raw).digest() != md5:
            raise RuntimeError("Failed to load audio: FLAC stream does not match its MD5 signature")
    scale = float(1 << (info.bits_per_sample - 1))
    return (pcm.astype(np.float64) / scale).astype(np.float32), int(info.sample_rate)


def read_pcm_file(source: Union[str, bytes]) -> Tuple[np.ndarray, int]:
    """the containers decoded without ffmpeg: RIFF/WAVE and native FLAC -> (f32 [frames, channels], sample rate)"""
    return read_wav(source) if is_wav(source) else read_flac(source)


def to_s16(x: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(np.asarray(x, np.float64) * 32768.0), -32768, 32767).astype("<i2")


# ---------------------------------------------------------------------------------------------------------- resample
def _sinc_kernel(orig: int, new: int, zeros: int = 6, rolloff: float = 0.99) -> Tuple[torch.Tensor, int]:
    """Filter bank of the rational resampler ``orig -> new`` (both already divided by their gcd): row p holds the taps
    that produce output phase p from ``2 * width + orig`` consecutive input samples.  f32 [new, 1, 2*width+orig]."""
    cutoff = min(orig, new) * rolloff
    width = math.ceil(zeros * orig / cutoff)
    taps = torch.arange(-width, width + orig, dtype=torch.float64)[None, :] / orig
    phase = -torch.arange(new, dtype=torch.float64)[:, None] / new
    t = ((phase + taps) * cutoff).clamp_(-zeros, zeros)
    window = torch.cos(t * (math.pi / zeros / 2)) ** 2
    t = t * math.pi
    sinc = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / torch.where(t == 0, torch.ones_like(t), t))
    return (sinc * window * (cutoff / orig)).to(torch.float32)[:, None, :], width


def resample(audio: Union[torch.Tensor, np.ndarray], in_sr: int, out_sr: int) -> torch.Tensor:
    """[..., n] at ``in_sr`` -> [..., ceil(n * out_sr / in_sr)] at ``out_sr``."""
    x = torch.as_tensor(audio)
    if not x.is_floating_point():
        x = x.float()
    if in_sr == out_sr or x.shape[-1] == 0:
        return x
    g = math.gcd(int(in_sr), int(out_sr))
    orig, new = int(in_sr) // g, int(out_sr) // g
    kernel, width = _sinc_kernel(orig, new)
    lead = x.shape[:-1]
    flat = x.reshape(-1, x.shape[-1])
    n = flat.shape[-1]
    padded = F.pad(flat, (width, width + orig))
    y = F.conv1d(padded[:, None], kernel.to(device=x.device, dtype=x.dtype), stride=orig)     # [b, new, blocks]
    y = y.transpose(1, 2).reshape(flat.shape[0], -1)[:, : math.ceil(new * n / orig)]
    return y.reshape(*lead, y.shape[-1])


def resample_blocks(blocks: Iterator[np.ndarray], in_sr: int, out_sr: int, block_periods: int = 4096) -> Iterator[np.ndarray]:
    """Streaming form of ``resample`` for mono f32 blocks of any sizes: yields the samples ``resample`` gives for the
    concatenated input (same taps per output sample; equal to f32 summation order, ~5e-7).  Input is regrouped into runs
    of ``block_periods * orig`` samples with a ``width`` halo."""
    if in_sr == out_sr:
        yield from blocks
        return
    g = math.gcd(int(in_sr), int(out_sr))
    orig, new = int(in_sr) // g, int(out_sr) // g
    kernel, width = _sinc_kernel(orig, new)
    taps = 2 * width + orig                        # conv window of one output period; periods start every `orig` samples
    run = block_periods * orig
    pending = np.zeros(width, np.float32)          # the left zero padding of the whole signal, then the unconsumed input
    total_in = emitted = 0

    def periods(buf: np.ndarray) -> np.ndarray:
        y = F.conv1d(torch.from_numpy(np.ascontiguousarray(buf))[None, None], kernel, stride=orig)
        return y.transpose(1, 2).reshape(-1).numpy()

    for blk in blocks:
        blk = np.asarray(blk, np.float32)
        total_in += len(blk)
        pending = np.concatenate([pending, blk])
        while len(pending) >= run - orig + taps:      # block_periods windows lie entirely inside real samples
            out = periods(pending[: run - orig + taps])
            emitted += len(out)
            yield out
            pending = pending[run:]
    tail = np.concatenate([pending, np.zeros(width + orig, np.float32)])
    left = math.ceil(new * total_in / orig) - emitted
    if left > 0 and len(tail) >= taps:
        yield periods(tail)[:left]


# ------------------------------------------------------------------------------------------------- voice-band filter
def _biquad(x: torch.Tensor, b: Tuple[float, float, float], a: Tuple[float, float, float]) -> torch.Tensor:
    from scipy.signal import lfilter
    y = lfilter(np.asarray(b, np.float64) / a[0], np.asarray(a, np.float64) / a[0], x.detach().cpu().numpy().astype(np.float64), axis=-1)
    return torch.from_numpy(y.astype(np.float32)).clamp_(-1.0, 1.0)


def lowpass_biquad(x: torch.Tensor, sr: int, cutoff: float, q: float = 0.707) -> torch.Tensor:
    w0 = 2 * math.pi * cutoff / sr
    alpha = math.sin(w0) / 2 / q
    c = math.cos(w0)
    return _biquad(x, ((1 - c) / 2, 1 - c, (1 - c) / 2), (1 + alpha, -2 * c, 1 - alpha))


def highpass_biquad(x: torch.Tensor, sr: int, cutoff: float, q: float = 0.707) -> torch.Tensor:
    w0 = 2 * math.pi * cutoff / sr
    alpha = math.sin(w0) / 2 / q
    c = math.cos(w0)
    return _biquad(x, ((1 + c) / 2, -1 - c, (1 + c) / 2), (1 + alpha, -2 * c, 1 - alpha))


def voice_freq_filter(wf: Union[torch.Tensor, np.ndarray], sr: int, upper_freq: Optional[int] = None,
                      lower_freq: Optional[int] = None) -> torch.Tensor:
    wf = torch.from_numpy(wf) if isinstance(wf, np.ndarray) else wf
    upper_freq = 5000 if upper_freq is None else upper_freq
    lower_freq = 200 if lower_freq is None else lower_freq
    assert upper_freq > lower_freq, f"upper_freq {upper_freq} must but greater than lower_freq {lower_freq}"
    return highpass_biquad(lowpass_biquad(wf, sr, upper_freq), sr, lower_freq)


# ---------------------------------------------------------------------------------------------------------- loading
def _ffmpeg_cmd(source: str, sr: int, mono: bool = True) -> List[str]:
    return ["ffmpeg", "-loglevel", "error", "-nostdin", "-threads", "0", "-i", source, "-f", "s16le",
            "-ac", "1" if mono else "2", "-acodec", "pcm_s16le", "-ar", str(sr), "-"]


def check_source(file: Union[str, bytes]):
    if isinstance(file, str) and "://" in file:
        raise NotImplementedError("URL sources need yt-dlp / network access -- out of scope (DESIGN.md section 7)")


def load_audio(file: Union[str, bytes], sr: int = SAMPLE_RATE, verbose: Optional[bool] = True, only_ffmpeg: bool = False,
               mono: bool = True) -> np.ndarray:
    """File path or file bytes -> f32 waveform at ``sr`` ([n] mono, [2, n] otherwise), values on the s16 grid like the
    reference's ``-f s16le`` pipe (audio/utils.py:63-125)."""
    check_source(file)
    if is_wav(file) or is_flac(file):
        x, in_sr = read_pcm_file(file)
        if mono:
            x = x.mean(axis=1, dtype=np.float64).astype(np.float32) if x.shape[1] > 1 else x[:, 0]
            y = resample(torch.from_numpy(np.ascontiguousarray(x)), in_sr, sr).numpy()
        else:
            x = x if x.shape[1] == 2 else np.repeat(x.mean(axis=1, keepdims=True), 2, axis=1)
            y = resample(torch.from_numpy(np.ascontiguousarray(x.T)), in_sr, sr).numpy()
        return to_s16(y).astype(np.float32) / 32768.0
    if shutil.which("ffmpeg") is None:
        raise RuntimeError("Failed to load audio: only RIFF/WAVE and FLAC sources can be decoded without ffmpeg on PATH")
    is_bytes = isinstance(file, (bytes, bytearray))
    try:
        out = subprocess.run(_ffmpeg_cmd("pipe:" if is_bytes else file, sr, mono), input=file if is_bytes else None,
                             capture_output=True, check=True).stdout
    except subprocess.CalledProcessError as e:
        raise RuntimeError(f"FFmpeg failed to load audio: {e.stderr.decode()}") from e
    wf = np.frombuffer(out, np.int16).flatten().astype(np.float32) / 32768.0
    return wf if mono else wf.reshape(-1, 2).transpose(1, 0)


def get_metadata(audiofile: Union[str, bytes, np.ndarray, torch.Tensor]) -> dict:
    """dict(sr, duration) -- audio/utils.py:152-182 (there parsed from ffmpeg's banner)."""
    if isinstance(audiofile, (np.ndarray, torch.Tensor)):
        return dict(sr=SAMPLE_RATE, duration=audiofile.shape[-1] / SAMPLE_RATE)
    if is_wav(audiofile):
        with _open_binary(audiofile) as f:
            info = _parse_wav(f)
        return dict(sr=info.sr, duration=info.n_frames / info.sr if info.sr else None)
    if is_flac(audiofile):
        fi = flac_info(audiofile)
        return dict(sr=fi["sr"], duration=(fi["frames"] / fi["sr"]) if fi["frames"] else None)
    if shutil.which("ffmpeg") is None:
        return dict(sr=None, duration=None)
    import re
    is_bytes = isinstance(audiofile, (bytes, bytearray))
    p = subprocess.run(["ffmpeg", "-hide_banner", "-i", "-" if is_bytes else audiofile],
                       input=audiofile if is_bytes else None, capture_output=True)
    text = p.stderr.decode(errors="ignore")
    sr = re.findall(r"\n.+Stream.+Audio.+\D+(\d+) Hz", text)
    dur = re.findall(r"Duration: ([\d:]+\.\d+),", text)
    duration = None
    if dur:
        h, m, s = dur[0].split(":")
        duration = int(h) * 3600 + int(m) * 60 + float(s)
    return dict(sr=int(sr[0]) if sr else None, duration=duration)


def get_samplerate(audiofile: Union[str, bytes]) -> Optional[int]:
    return get_metadata(audiofile).get("sr")


def audio_to_tensor_resample(audio, original_sample_rate: Optional[int] = None, target_sample_rates=None, **kwargs) -> torch.Tensor:
    """audio/utils.py:189-214"""
    if target_sample_rates and isinstance(target_sample_rates, int):
        target_sample_rates = [target_sample_rates]
    if isinstance(audio, (str, bytes)):
        if target_sample_rates:
            original_sample_rate = target_sample_rates[0]
        audio = load_audio(audio, sr=original_sample_rate or SAMPLE_RATE, **kwargs)
    elif not original_sample_rate:
        original_sample_rate = SAMPLE_RATE
    if isinstance(audio, np.ndarray):
        audio = torch.from_numpy(audio)
    audio = audio.float()
    if target_sample_rates and original_sample_rate not in target_sample_rates:
        audio = resample(audio, original_sample_rate, target_sample_rates[0])
    return audio


def reject_denoiser(denoiser, demucs=None):
    if denoiser or demucs:
        raise NotImplementedError("denoisers (demucs / dfnet / noisereduce) need their model files -- out of scope "
                                  "(DESIGN.md section 7)")


def prep_audio(audio: Union[str, np.ndarray, torch.Tensor, bytes], denoiser: Optional[str] = None,
               denoiser_options: Optional[dict] = None, only_voice_freq: bool = False, only_ffmpeg: bool = False,
               verbose: Optional[bool] = False, sr: Optional[int] = None, demucs=None, demucs_options=None) -> torch.Tensor:
    """Any supported input -> mono waveform tensor (audio/__init__.py:74-149).  Arrays and tensors are taken as already
    sampled at ``sr`` and are returned as they are (same object, same device) unless ``only_voice_freq``."""
    reject_denoiser(denoiser, demucs)
    sr = sr or SAMPLE_RATE
    if isinstance(audio, (str, bytes)):
        audio = torch.from_numpy(load_audio(audio, sr=sr, verbose=verbose, only_ffmpeg=only_ffmpeg))
    elif isinstance(audio, np.ndarray):
        audio = torch.from_numpy(audio)
    if only_voice_freq:
        audio = voice_freq_filter(audio.cpu(), sr)
    return audio


# ------------------------------------------------------------------------------------------------- streamed PCM source
class PcmStream:
    """s16le mono bytes at the target rate, pulled on demand: the role of the reference's ffmpeg child process
    (audio/__init__.py:552-591).  ``read(n_bytes)`` returns fewer bytes only at the end of the source."""

    def __init__(self, chunks: Iterator[bytes], closer: Optional[Callable[[], None]] = None):
        self._chunks, self._closer, self._left, self.ended = chunks, closer, b"", False

    @property
    def exhausted(self) -> bool:
        return self.ended and not self._left

    def read(self, n_bytes: int) -> bytes:
        parts, have = [self._left], len(self._left)
        while have < n_bytes and not self.ended:
            try:
                c = next(self._chunks)
            except StopIteration:
                self.ended = True
                break
            parts.append(c)
            have += len(c)
        data = b"".join(parts)
        self._left = data[n_bytes:]
        return data[:n_bytes]

    def close(self):
        self.ended = True
        if self._closer is not None:
            self._closer()
            self._closer = None


def _wav_pcm_stream(source: Union[str, bytes], sr: int, frames_per_read: int = 1 << 18) -> PcmStream:
    f = _open_binary(source)
    info = _parse_wav(f)

    def mono_blocks():
        left = info.n_frames
        while left > 0:
            raw = f.read(min(left, frames_per_read) * info.block)
            k = len(raw) // info.block
            if k == 0:
                return
            left -= k
            x = _decode_frames(raw[: k * info.block], info)
            yield x.mean(axis=1, dtype=np.float64).astype(np.float32) if x.shape[1] > 1 else np.ascontiguousarray(x[:, 0])

    return PcmStream((to_s16(b).tobytes() for b in resample_blocks(mono_blocks(), info.sr, sr)), f.close)


def _flac_pcm_stream(source: Union[str, bytes], sr: int, frames_per_read: int = 1 << 18) -> PcmStream:
    """a FLAC source as a pulled s16le stream: the file is decoded whole (a frame cannot be located without parsing the ones
    before it, and decoding runs at several thousand times real time), then handed out block-wise through the streaming resampler"""
    x, in_sr = read_flac(source)
    mono = x.mean(axis=1, dtype=np.float64).astype(np.float32) if x.shape[1] > 1 else np.ascontiguousarray(x[:, 0])

    def mono_blocks():
        for a in range(0, len(mono), frames_per_read):
            yield mono[a: a + frames_per_read]

    return PcmStream((to_s16(b).tobytes() for b in resample_blocks(mono_blocks(), in_sr, sr)))


def _ffmpeg_pcm_stream(source: str, sr: int) -> PcmStream:
    try:
        p = subprocess.Popen(_ffmpeg_cmd(source, sr), stdout=subprocess.PIPE)
    except (subprocess.SubprocessError, OSError) as e:
        raise RuntimeError(f"Failed to load audio: {e}") from e

    def chunks():
        while True:
            b = p.stdout.read(1 << 16)
            if not b:
                return
            yield b

    def close():
        if p.poll() is None:
            p.terminate()

    return PcmStream(chunks(), close)


def open_pcm_stream(source: Union[str, bytes], sr: int) -> PcmStream:
    check_source(source)
    if is_wav(source):
        return _wav_pcm_stream(source, sr)
    if is_flac(source):
        return _flac_pcm_stream(source, sr)
    if isinstance(source, str) and shutil.which("ffmpeg") is not None:
        return _ffmpeg_pcm_stream(source, sr)
    raise RuntimeError(f'FFmpeg failed to read "{source}".' if isinstance(source, str) else "Failed to load audio: "
                       "only RIFF/WAVE and FLAC sources can be decoded without ffmpeg on PATH")


# ------------------------------------------------------------------------------------------------------- AudioLoader
Section = Tuple[Optional[float], Optional[float]]


class AudioLoader:
    """Buffered, seekable view of an audio source in samples at ``sr`` (audio/__init__.py:152-638).

    In-memory sources (arrays, tensors, and files when ``stream`` is false) are prepared once on the first request;
    streamed sources are pulled from a PCM stream as the seek advances and can only move forward.  ``next_chunk(seek,
    size)`` returns up to ``size`` (default ``buffer_size``) samples starting at ``seek`` or ``None`` past the end;
    ``next_valid_chunk`` additionally confines the request to ``load_sections`` (pairs of seconds)."""

    def __init__(self, source: Union[str, np.ndarray, torch.Tensor, bytes], buffer_size: Union[int, str, None] = None,
                 stream: Optional[bool] = None, sr: Optional[int] = None, test_first_chunk: bool = True,
                 verbose: Optional[bool] = False, only_ffmpeg: bool = False, new_chunk_divisor: Optional[int] = 512,
                 save_path: Optional[str] = None, post_prep_callback: Optional[Callable] = None,
                 denoiser: Optional[str] = None, denoiser_options: Optional[dict] = None, only_voice_freq: bool = False,
                 demucs=None, demucs_options=None, load_sections: Optional[List[Section]] = None, negate_load: bool = False):
        if stream and not isinstance(source, str):
            raise NotImplementedError(f"``stream=True`` only supported for string ``source`` but got {type(source)}.")
        reject_denoiser(denoiser, demucs)
        self.source = source
        self._sr = sr or SAMPLE_RATE
        self.load_sections = self.negate_ts_sections(load_sections) if (negate_load and load_sections) else load_sections
        self._curr_load_section_index = -1
        self._curr_load_section_seeks: Tuple[Optional[int], Optional[int]] = (0, 0)
        self._buffer_size = self._valid_buffer_size(self.parse_chunk_size(self._sr * 30 if buffer_size is None else buffer_size))
        self._stream = isinstance(source, str) if stream is None else stream
        self._accum_samples = 0
        self.verbose = verbose
        self.only_ffmpeg = only_ffmpeg
        self.new_chunk_divisor = new_chunk_divisor
        self._post_prep_callback = post_prep_callback
        self._denoiser, self._denoiser_options = None, (denoiser_options or {})
        self._final_save_path = save_path
        self._only_voice_freq = only_voice_freq
        self._final_samples_to_save: List[torch.Tensor] = []
        meta = get_metadata(source)
        self._source_sr, self._duration_estimation = meta["sr"] or 0, meta["duration"] or 0
        self._total_sample_estimation = round(self._duration_estimation * self._sr)
        self._prev_seek: Optional[int] = None
        self._buffered_samples = torch.tensor([])
        self._pcm: Optional[PcmStream] = open_pcm_stream(source, self._sr) if (self._stream and isinstance(source, str)) else None
        if test_first_chunk and self.next_chunk(0) is None:
            raise RuntimeError(f'FFmpeg failed to read "{source}".' if isinstance(source, str) else "Failed to load audio.")

    # -- properties the window loop reads
    @property
    def buffer_size(self) -> int:
        return self._buffer_size

    @buffer_size.setter
    def buffer_size(self, size: int):
        self._buffer_size = self._valid_buffer_size(size)

    @property
    def sr(self) -> int:
        return self._sr

    @property
    def source_sr(self) -> int:
        return self._source_sr

    @property
    def stream(self) -> bool:
        return self._stream

    @property
    def prev_seek(self) -> Optional[int]:
        return self._prev_seek

    @property
    def curr_load_section_index(self) -> int:
        return self._curr_load_section_index

    @property
    def curr_load_section_seeks(self) -> Tuple[Optional[int], Optional[int]]:
        return self._curr_load_section_seeks

    @staticmethod
    def _valid_buffer_size(size: int) -> int:
        if size < 0:
            raise ValueError("buffer size must be at least 0")
        return size

    @staticmethod
    def negate_ts_sections(ts_sections: List[Section]) -> List[Section]:
        """the complement of a sorted section list: gaps between sections plus the open head and tail"""
        out = [(0.0, ts_sections[0][0])]
        out += [(a[1], b[0]) for a, b in zip(ts_sections[:-1], ts_sections[1:])]
        out.append((ts_sections[-1][1], None))
        return [s for s in out if s[0] != s[1]]

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.terminate()

    def __del__(self):
        self.terminate()

    def parse_chunk_size(self, chunk_size: Union[int, str]) -> int:
        if isinstance(chunk_size, int):
            return chunk_size
        if not chunk_size.endswith("s"):
            raise ValueError('string ``chunk_size`` must end with "s"')
        return round(float(chunk_size[:-1]) * self._sr)

    def get_duration(self, ndigits: Optional[int] = None) -> float:
        dur = self._duration_estimation
        if self._stream:
            seen = (self._accum_samples or 0) / self._sr
            dur = self._duration_estimation if seen < self._duration_estimation else seen
        return dur if ndigits is None else round(dur, ndigits=ndigits)

    def get_total_samples(self) -> int:
        if not self._stream:
            return self._total_sample_estimation
        if (self._accum_samples / self._sr) < self._duration_estimation:
            return self._total_sample_estimation
        return self._accum_samples

    def update_post_prep_callback(self, callback: Optional[Callable]):
        self._post_prep_callback = callback
        if callback is not None and len(self._buffered_samples):
            callback(self._buffered_samples)

    def divisible_min_chunk(self, min_chunk: int) -> int:
        d = self.new_chunk_divisor
        if d and min_chunk % d:
            return min_chunk + d - min_chunk % d
        return min_chunk

    # -- buffer mechanics
    def _prep(self, audio) -> torch.Tensor:
        if self._stream:                                    # per block of freshly converted samples
            audio = torch.from_numpy(audio)
            return voice_freq_filter(audio.cpu(), self._sr) if self._only_voice_freq else audio
        return prep_audio(audio, only_voice_freq=self._only_voice_freq, only_ffmpeg=self.only_ffmpeg,
                          verbose=self.verbose, sr=self._sr)

    def _seek_buffered_samples(self, seek: int) -> int:
        """moves the buffer start to ``seek``; returns how many samples must be read and dropped before it"""
        if self._prev_seek is None:
            if self._pcm is None:
                self._buffered_samples = self._prep(self.source)
                if self._final_save_path:
                    self._final_samples_to_save.append(self._buffered_samples.cpu())
                self._total_sample_estimation = self._buffered_samples.shape[-1]
                self._duration_estimation = self._total_sample_estimation / self._sr
                self._buffered_samples = self._buffered_samples[seek:]
                skip = 0
            else:
                self._buffered_samples = torch.tensor([])
                skip = seek
        else:
            assert seek >= self._prev_seek, "``seek`` must be >= the previous ``seek`` value"
            delta = seek - self._prev_seek
            skip = max(0, delta - len(self._buffered_samples))
            self._buffered_samples = self._buffered_samples[delta:]
        self._prev_seek = seek
        return skip

    def _read_samples(self, samples: int) -> bytes:
        if self._pcm is None or self._pcm.exhausted:
            return b""
        return self._pcm.read(samples * 2)

    def _prep_samples(self, new_bytes: bytes, samples_to_discard: Optional[int] = None) -> torch.Tensor:
        if samples_to_discard:
            assert not len(self._buffered_samples)
            new_bytes = new_bytes[samples_to_discard * 2:]
        new = np.frombuffer(new_bytes[: len(new_bytes) // 2 * 2], np.int16).flatten().astype(np.float32) / 32768.0
        self._accum_samples += new.shape[-1]
        prepped = self._prep(new)
        if self._final_save_path:
            self._final_samples_to_save.append(prepped.cpu())
        if self._post_prep_callback is not None:
            self._post_prep_callback(prepped)
        return prepped

    def _read_append_to_buffer(self, samples_to_read: int, samples_to_discard: Optional[int] = None):
        data = self._read_samples(samples_to_read)
        if not data:
            return
        new = self._prep_samples(data, samples_to_discard)
        self._buffered_samples = torch.concat([self._buffered_samples, new], dim=-1) if len(self._buffered_samples) else new

    def next_chunk(self, seek: int, size: Optional[int] = None) -> Optional[torch.Tensor]:
        skip = self._seek_buffered_samples(seek)
        keep = max(self._buffer_size, size or 0) - len(self._buffered_samples)
        if keep > 0:
            keep = self.divisible_min_chunk(keep)
        self._read_append_to_buffer(max(skip + keep, 0), skip)
        samples = self._buffered_samples[: self._buffer_size if size is None else size]
        return samples if len(samples) else None

    def next_valid_chunk(self, seek: int, size: Optional[int] = None) -> Tuple[Optional[torch.Tensor], int]:
        if not self.load_sections:
            return self.next_chunk(seek, size=size), seek
        while (stop := self._curr_load_section_seeks[1]) is not None and seek + 1 >= stop:
            if not self.skip_to_next_section():
                return None, seek
            if seek < self._curr_load_section_seeks[0]:
                seek = self._curr_load_section_seeks[0]
        chunk = self.next_chunk(seek, size=size)
        if chunk is None:
            return None, seek
        stop = self._curr_load_section_seeks[1]
        if stop is not None and seek + chunk.size(-1) > stop:
            chunk = chunk[..., : stop - seek]
        return chunk, seek

    def skip_to_next_section(self) -> bool:
        if not self.load_sections or self._curr_load_section_index + 1 >= len(self.load_sections):
            return False
        self._curr_load_section_index += 1
        start, end = self.load_sections[self._curr_load_section_index]
        self._curr_load_section_seeks = (None if start is None else round(start * self._sr),
                                         None if end is None else round(end * self._sr))
        return True

    # -- bookkeeping
    def save_final_audio(self, path: Optional[str] = None):
        if not self._final_samples_to_save:
            warnings.warn("Failed to save final audio. No stored final audio samples found.", stacklevel=2)
            return
        if not (path or self._final_save_path):
            warnings.warn("Failed to save denoised audio. No specified path to save.", stacklevel=2)
            return
        write_wav(path or self._final_save_path, torch.cat(self._final_samples_to_save), self._sr)

    def terminate(self):
        pcm = getattr(self, "_pcm", None)
        if pcm is not None:
            pcm.close()
        if getattr(self, "_final_save_path", None) and getattr(self, "_final_samples_to_save", None):
            self.save_final_audio()
            self._final_samples_to_save = []

    def validate_external_args(self, sr=None, vad=None, stream=None, denoiser=None, denoiser_options=None,
                               only_voice_freq=False):
        if sr and sr != self._sr:
            raise ValueError(f"AudioLoader must be initialized with ``sr={sr}`` but ``sr`` of this instance is {self._sr}.")
        if vad:
            raise NotImplementedError("vad=True needs the Silero model (torch.hub, network) -- out of scope offline")
        if stream and not self._stream:
            warnings.warn("``stream=True`` will have no effect unless specified at AudioLoader initialization.", stacklevel=2)
        reject_denoiser(denoiser)
        if only_voice_freq and not self._only_voice_freq:
            warnings.warn("``only_voice_freq=True`` will have no effect unless specified at AudioLoader initialization.",
                          stacklevel=2)
        return self._stream, self._denoiser, self._denoiser_options, self._only_voice_freq


def audioloader_not_supported(audio):
    if isinstance(audio, AudioLoader):
        raise NotImplementedError("This function does not support AudioLoader instances.")
