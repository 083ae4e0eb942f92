"""ORACLE stand-in package for the dependency ``openai-whisper==20250625`` (see oracle/__init__.py)."""
__version__ = "20250625"

from . import audio, decoding, model, timing, tokenizer  # noqa: F401
from .audio import log_mel_spectrogram, pad_or_trim  # noqa: F401
from .decoding import DecodingOptions, DecodingResult, decode, detect_language  # noqa: F401
from .model import ModelDimensions, Whisper, build_model, dims_for  # noqa: F401


def available_models():
    return list(model._DIMS.keys())


def load_model(name: str, device=None, download_root=None, in_memory: bool = False, **kw):
    """No checkpoints exist offline: returns the architecture `name` with seeded random weights."""
    m = build_model(name, seed=kw.pop("seed", 1234), std=kw.pop("std", 0.02))
    return m.to(device) if device is not None else m


def transcribe(*a, **k):  # pragma: no cover - only so that `whisper.transcribe` resolves
    raise NotImplementedError("oracle stand-in: use stable_whisper's transcribe on top of this package")
