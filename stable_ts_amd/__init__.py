"""stable_ts_amd -- MI355X-native Whisper transcription + word-level alignment (stable-ts hot path)."""
__version__ = "0.1.0"
