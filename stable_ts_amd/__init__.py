"""stable_ts_amd -- MI355X-native Whisper transcription + word-level alignment (the stable-ts hot path).

    import stable_ts_amd as stable_whisper
    model = stable_whisper.load_model('large-v3')          # or weights='random' offline
    result = model.transcribe(audio)                        # WhisperResult with word timestamps
    result = model.align(audio, text, language='en')

The arithmetic runs in libswx.so (hand-written gfx950 HIP kernels, include/swx.h); importing the package does not
need a GPU, using a model does, and there is no CPU fallback.
"""
__version__ = "0.1.0"

from .model import BENCH_WEIGHTS, Whisper, available_models, dims_for, load_model, random_state_dict  # noqa: F401
from .engine import Engine, ModelDimensions  # noqa: F401
from .result import Segment, WhisperResult, WordTiming  # noqa: F401
from .decoding import DecodingOptions, DecodingResult  # noqa: F401
from .transcribe import transcribe_stable  # noqa: F401
from .audio_io import AudioLoader, load_audio, prep_audio  # noqa: F401
from .alignment import align, align_words, refine  # noqa: F401
from .locator import locate  # noqa: F401
from .spans import plan_spans, transcribe_spans  # noqa: F401
from .non_whisper import transcribe_any  # noqa: F401
from .audio import (  # noqa: F401
    SAMPLE_RATE, N_FFT, HOP_LENGTH, CHUNK_LENGTH, N_SAMPLES, N_FRAMES, N_SAMPLES_PER_TOKEN, FRAMES_PER_SECOND,
    TOKENS_PER_SECOND, pad_or_trim,
)
