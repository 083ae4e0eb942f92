"""Host driver of the on-device decoding loop.

Mirrors the surface the reference consumes from ``whisper.decoding`` + ``stable_whisper/decode.py``:
``DecodingOptions`` / ``DecodingResult`` and ``decode_stable(model, mel, options, ts_token_mask, audio_features)``
(decode.py:70-110).  Everything per-step (decoder forward, logit filters, greedy / beam update) runs inside
``swx_decode``; this module only prepares the initial tokens and turns the candidate sequences that come back into a
``DecodingResult`` (ranking, text, avg_logprob, compression ratio -- upstream ``DecodingTask.run`` tail).
"""
import zlib
from dataclasses import dataclass, field, replace
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from .audio import CHUNK_LENGTH
from .tokenizer import Tokenizer, get_tokenizer


def compression_ratio(text: str) -> float:
    raw = text.encode("utf-8")
    return len(raw) / len(zlib.compress(raw))


@dataclass(frozen=True)
class DecodingOptions:
    task: str = "transcribe"
    language: Optional[str] = None
    temperature: float = 0.0
    sample_len: Optional[int] = None
    best_of: Optional[int] = None
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    length_penalty: Optional[float] = None
    prompt: Optional[Union[str, List[int]]] = None
    prefix: Optional[Union[str, List[int]]] = None
    suppress_tokens: Optional[Union[str, Iterable[int]]] = "-1"
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    fp16: bool = True
    # extension (not in upstream): EOT is suppressed until this many tokens were sampled -- used only to give
    # random-weight benchmarks a fixed decode length; 0 keeps the reference behaviour
    min_tokens: int = 0
    seed: int = 0


@dataclass(frozen=True)
class DecodingResult:
    audio_features: object
    language: str
    language_probs: Optional[Dict[str, float]] = None
    tokens: List[int] = field(default_factory=list)
    text: str = ""
    avg_logprob: float = np.nan
    no_speech_prob: float = np.nan
    temperature: float = np.nan
    compression_ratio: float = np.nan


class DecodingPlan:
    """The host-side half of upstream's DecodingTask.__init__ (option checks, initial tokens, suppress list)."""

    def __init__(self, model, options: DecodingOptions):
        self.model = model
        self.options = options
        if options.beam_size is not None and options.best_of is not None:
            raise ValueError("beam_size and best_of can't be given together")
        if options.temperature == 0 and options.best_of is not None:
            raise ValueError("best_of with greedy sampling (T=0) is not compatible")
        if options.patience is not None and options.beam_size is None:
            raise ValueError("patience requires beam_size to be given")
        if options.length_penalty is not None and not (0 <= options.length_penalty <= 1):
            raise ValueError("length_penalty (alpha) should be a value between 0 and 1")
        language = options.language or "en"
        self.tokenizer: Tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages,
                                                  language=language, task=options.task)
        tok = self.tokenizer
        self.n_group = options.beam_size or options.best_of or 1
        self.n_ctx = model.dims.n_text_ctx
        self.sample_len = options.sample_len or model.dims.n_text_ctx // 2
        self.sot_sequence = tok.sot_sequence_including_notimestamps if options.without_timestamps else tok.sot_sequence
        self.initial_tokens = self._initial_tokens()
        self.sample_begin = len(self.initial_tokens)
        self.sot_index = self.initial_tokens.index(tok.sot)
        self.suppress = self._suppress_tokens() if options.suppress_tokens else ()
        self.max_initial_timestamp_index = None
        if not options.without_timestamps and options.max_initial_timestamp:
            precision = CHUNK_LENGTH / model.dims.n_audio_ctx
            self.max_initial_timestamp_index = round(options.max_initial_timestamp / precision)

    def _initial_tokens(self) -> Tuple[int, ...]:
        tok, o = self.tokenizer, self.options
        tokens = list(self.sot_sequence)
        if o.prefix:
            p = tok.encode(" " + o.prefix.strip()) if isinstance(o.prefix, str) else list(o.prefix)
            if self.sample_len is not None:
                p = p[-(self.n_ctx // 2 - self.sample_len):]
            tokens = tokens + p
        if o.prompt:
            p = tok.encode(" " + o.prompt.strip()) if isinstance(o.prompt, str) else list(o.prompt)
            tokens = [tok.sot_prev] + p[-(self.n_ctx // 2 - 1):] + tokens
        return tuple(tokens)

    def _suppress_tokens(self) -> Tuple[int, ...]:
        tok = self.tokenizer
        s = self.options.suppress_tokens
        if isinstance(s, str):
            s = [int(t) for t in s.split(",")]
        s = list(s)
        if -1 in s:
            s = [t for t in s if t >= 0]
            s.extend(tok.non_speech_tokens)
        s.extend([tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm])
        if tok.no_speech is not None:
            s.append(tok.no_speech)
        return tuple(sorted(set(s)))

    def engine_kwargs(self) -> dict:
        tok, o = self.tokenizer, self.options
        blank = tok.encode(" ")
        return dict(
            n_group=self.n_group, beam=o.beam_size is not None, temperature=o.temperature, patience=o.patience,
            sample_len=self.sample_len, sot_index=self.sot_index, suppress_blank=o.suppress_blank,
            apply_timestamp_rules=not o.without_timestamps, max_initial_timestamp_index=self.max_initial_timestamp_index,
            eot=tok.eot, sot=tok.sot, no_timestamps=tok.no_timestamps, timestamp_begin=tok.timestamp_begin,
            no_speech=tok.no_speech if tok.no_speech is not None else -1, blank_token=blank[0] if blank else -1,
            suppress_tokens=self.suppress, min_tokens=o.min_tokens, seed=o.seed)

    def results(self, out: dict, audio_features: Sequence, languages: Sequence[str]) -> List[DecodingResult]:
        """upstream DecodingTask.run tail: cut at EOT, rank the group, decode text, statistics."""
        tok, o = self.tokenizer, self.options
        sb = out["sample_begin"]
        res = []
        for w in range(out["tokens"].shape[0]):
            cands, slps = [], []
            for k in range(out["tokens"].shape[1]):
                ln = int(out["lens"][w, k])
                if ln < 0:
                    continue
                cands.append(out["tokens"][w, k, sb: sb + ln].tolist())
                slps.append(float(out["sum_logprobs"][w, k]))
            scores = []
            for t, lp in zip(cands, slps):
                length = len(t)
                penalty = length if o.length_penalty is None else ((5 + length) / 6) ** o.length_penalty
                with np.errstate(divide="ignore", invalid="ignore"):
                    scores.append(np.float64(lp) / penalty)
            best = int(np.argmax(scores))
            tokens = cands[best]
            text = tok.decode(tokens).strip()
            res.append(DecodingResult(
                audio_features=audio_features[w], language=languages[w], tokens=tokens, text=text,
                avg_logprob=slps[best] / (len(tokens) + 1), no_speech_prob=float(out["no_speech_prob"][w]),
                temperature=o.temperature, compression_ratio=compression_ratio(text)))
        return res


def decode_windows(model, xkv, options: DecodingOptions, ts_token_mask=None, prompts: Optional[Sequence[Sequence[int]]] = None,
                   audio_features=None) -> List[DecodingResult]:
    """Decode W windows (the batch inside `xkv`) in lockstep.  `prompts` optionally gives per-window prompt token lists;
    windows whose initial-token length differs are decoded in separate jobs (each job is one lockstep batch)."""
    W = xkv.n_windows
    if prompts is None:
        prompts = [options.prompt] * W
    plans = [DecodingPlan(model, replace(options, prompt=(list(p) if p else None))) for p in prompts]
    if options.language is None and model.is_multilingual:
        raise ValueError("language must be resolved (detect_language) before decode_windows")
    languages = [options.language or "en"] * W
    out_all = [None] * W
    if audio_features is None:
        audio_features = [None] * W
    # group windows by initial length; windows of one group must be contiguous in xkv -> run per window otherwise
    same = len({p.sample_begin for p in plans}) == 1
    if same:
        masks = None
        if ts_token_mask is not None:
            masks = ts_token_mask if ts_token_mask.ndim == 2 else ts_token_mask[None].expand(W, -1)
        out = model.engine.decode(xkv, [list(p.initial_tokens) for p in plans], ts_mask=masks, **plans[0].engine_kwargs())
        return plans[0].results(out, audio_features, languages)
    raise NotImplementedError("windows with different prompt lengths must be decoded as separate jobs "
                              "(use model.engine.cross_kv per window)")
