"""ORACLE (test infrastructure). ``whisper.tokenizer`` stand-in.

The tokenizer is host-side text bookkeeping, not hot-path arithmetic, and no real
vocabulary exists offline; the oracle and the product therefore share ONE synthetic
vocabulary definition (stable_ts_amd/tokenizer.py) so that token ids mean the same
thing on both sides.  Nothing here is measured or checked for parity.
"""
from stable_ts_amd.tokenizer import (  # noqa: F401
    LANGUAGES, TO_LANGUAGE_CODE, SyntheticEncoding, Tokenizer, get_encoding, get_tokenizer,
)
