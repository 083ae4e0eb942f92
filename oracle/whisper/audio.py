"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of ``whisper.audio`` from the third-party dependency
``openai-whisper==20250625`` (pinned by /root/reference setup.py:30 and
stable_whisper/whisper_compatibility.py:11-21).  That package is absent from
/root/reference and from this image, so its published algorithm is restated
here; the reference call sites are
``whisper_word_level/original_whisper.py:528-530`` and ``alignment.py:410-413``.

Pinning: `mel_filters` / `log_mel_spectrogram` are checked against
``transformers.WhisperFeatureExtractor`` (an independent implementation that is
installed in this image) in ``tests/test_oracle_pinning.py``.
"""
import math
from functools import lru_cache

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE  # 480000
N_FRAMES = N_SAMPLES // HOP_LENGTH  # 3000
N_SAMPLES_PER_TOKEN = HOP_LENGTH * 2  # 320
FRAMES_PER_SECOND = SAMPLE_RATE // HOP_LENGTH  # 100
TOKENS_PER_SECOND = SAMPLE_RATE // N_SAMPLES_PER_TOKEN  # 50


def pad_or_trim(array, length: int = N_SAMPLES, *, axis: int = -1):
    """Zero-pad on the right or cut to ``length`` along ``axis``."""
    if torch.is_tensor(array):
        if array.shape[axis] > length:
            array = array.index_select(dim=axis, index=torch.arange(length, device=array.device))
        if array.shape[axis] < length:
            pad_widths = [(0, 0)] * array.ndim
            pad_widths[axis] = (0, length - array.shape[axis])
            array = F.pad(array, [p for sizes in pad_widths[::-1] for p in sizes])
    else:
        if array.shape[axis] > length:
            array = array.take(indices=range(length), axis=axis)
        if array.shape[axis] < length:
            pad_widths = [(0, 0)] * array.ndim
            pad_widths[axis] = (0, length - array.shape[axis])
            array = np.pad(array, pad_widths)
    return array


def _hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


@lru_cache(maxsize=None)
def _mel_filters_np(n_mels: int) -> np.ndarray:
    """librosa.filters.mel(sr=16000, n_fft=400, n_mels=n_mels) (Slaney scale, Slaney norm), f32 [n_mels, 201].

    upstream ships this matrix as assets/mel_filters.npz (generated with librosa);
    the asset is not available offline so it is regenerated from librosa's formula.
    """
    n_freqs = N_FFT // 2 + 1
    fftfreqs = np.linspace(0.0, SAMPLE_RATE / 2.0, n_freqs)
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(SAMPLE_RATE / 2.0), n_mels + 2)
    mel_f = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, n_freqs), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


def mel_filters(device, n_mels: int) -> torch.Tensor:
    assert n_mels in {80, 128}, f"Unsupported n_mels: {n_mels}"
    return torch.from_numpy(_mel_filters_np(n_mels)).to(device)


def log_mel_spectrogram(audio, n_mels: int = 80, padding: int = 0, device=None) -> torch.Tensor:
    """f32 [..., n_mels, n_frames] log-mel spectrogram (upstream whisper/audio.py::log_mel_spectrogram)."""
    if not torch.is_tensor(audio):
        audio = torch.from_numpy(np.asarray(audio))
    if device is not None:
        audio = audio.to(device)
    audio = audio.to(torch.float32)
    if padding > 0:
        audio = F.pad(audio, (0, padding))
    window = torch.hann_window(N_FFT).to(audio.device)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    filters = mel_filters(audio.device, n_mels)
    mel_spec = filters @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    return log_spec
