"""Import stub so that /root/reference's stable_whisper imports in this container (no torchaudio here).
Only used by tests/golden/make_golden.py; the audio I/O paths that need torchaudio are never called."""
from . import functional  # noqa: F401
