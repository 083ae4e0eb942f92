"""Span-parallel transcription: the reference's sequential algorithm on several contiguous spans of one recording at once.

SURVEY.md section 8e: window k+1 of ``transcribe()`` depends on window k (seek from the last timestamp token, prompt
from the previous text; original_whisper.py:483-485, 533, 629-633, 680-682, 703-708), so the exact algorithm does not
shard by window.  It does shard by *span*: cut the recording at quiet places into contiguous spans, run the sequential
algorithm on every span independently (fresh prompt at each span start) and concatenate the results shifted by their
span offsets.  The oracle for this mode is exact by construction -- the reference's ``transcribe()`` run once per span
and concatenated (``tests/test_spans_cpu.py``) -- unlike ``batch_size=N`` (fixed 30-s stride, no prompt carry-over).
The equality is for deterministic decoding (greedy / beam at temperature 0, thresholds that do not trigger the sampling
fallback); sampled fallbacks draw random numbers, and the spans' draws interleave here.

On one GPU the spans advance in lockstep: every device batch holds the current window of each live span, so the encoder,
the decode loop and the scoring pass run at batch = number of spans while each span still sees exactly the windows and
prompts the sequential algorithm produces.  Across GPUs the spans are dealt to the ranks (``parallel.transcribe_sharded(
..., mode="spans")``); nothing is exchanged until the final gather of the segment records.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .stabilization import host_single_thread
from .audio import N_SAMPLES, SAMPLE_RATE
from .result import WhisperResult
from .stabilization import NonSpeechPredictor


def plan_spans(audio: torch.Tensor, n_spans: int, search: float = 10.0, q_levels: int = 20, k_size: int = 5) -> List[Tuple[int, int]]:
    """Cuts ``audio`` (1-D, 16 kHz) into ``n_spans`` contiguous ``(start, end)`` sample ranges of about equal length.
    Each interior cut is moved to the middle of the longest non-speech section within ``search`` seconds of its nominal
    place (the reference's own loudness-based detector, stabilization/__init__.py:241-254, is the judge of "quiet");
    with no quiet place nearby the nominal cut is kept.  Spans shorter than one window are not produced."""
    total = int(audio.shape[-1])
    n_spans = max(1, min(int(n_spans), max(1, total // N_SAMPLES)))
    cuts = [0]
    half = int(search * SAMPLE_RATE)
    for k in range(1, n_spans):
        nominal = round(k * total / n_spans)
        lo, hi = max(cuts[-1] + N_SAMPLES // 2, nominal - half), min(total, nominal + half)
        cut = min(max(nominal, lo), hi)
        if hi - lo > SAMPLE_RATE // 10:
            pred = NonSpeechPredictor(q_levels=q_levels, k_size=k_size).predict(audio[lo:hi], offset=lo / SAMPLE_RATE)
            t = pred["timings"]
            if t is not None and len(t[0]):
                j = int(np.argmax(t[1] - t[0]))
                cut = int(round((t[0][j] + t[1][j]) / 2 * SAMPLE_RATE))
        if cut <= cuts[-1] or cut >= total:
            continue
        cuts.append(cut)
    cuts.append(total)
    return [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def merge_span_results(parts: Sequence[Tuple[int, WhisperResult]], language: Optional[str] = None) -> WhisperResult:
    """[(span start in samples, result of that span)] -> one result in recording time (the concatenation the reference's
    users build with ``offset_time`` + appending segments)."""
    segments, sections, texts = [], [], []
    for start, res in sorted(parts, key=lambda p: p[0]):
        off = start / SAMPLE_RATE
        res.offset_time(off)
        segments.extend(s.to_dict() for s in res.segments)
        sections.extend(dict(start=d["start"] + off, end=d["end"] + off) for d in res.nonspeech_sections)
        texts.append(res.text)
        language = language or res.language
    out = WhisperResult(dict(text="".join(texts), segments=segments, language=language), check_sorted=False)
    out.nonspeech_sections = sections
    return out


@host_single_thread
def transcribe_spans(model, audio, n_spans: int = 8, *, spans: Optional[List[Tuple[int, int]]] = None, search: float = 10.0,
                     **kw) -> WhisperResult:
    """``model.transcribe`` semantics per span, all spans advanced together on this device.  ``spans`` (sample ranges)
    overrides the automatic plan.  Options are those of ``transcribe`` except ``batch_size`` and ``clip_timestamps``."""
    from .transcribe import as_waveform, transcribe_stable
    for k in ("batch_size", "clip_timestamps"):
        if kw.get(k):
            raise NotImplementedError(f"{k} does not combine with span-parallel transcription")
    wave = as_waveform(audio, only_voice_freq=bool(kw.pop("only_voice_freq", False)))
    if wave.shape[-1] == 0:
        raise RuntimeError("Failed to load audio.")
    if spans is None:
        spans = plan_spans(wave.detach().float().cpu(), n_spans, search, kw.get("q_levels", 20), kw.get("k_size", 5))
    parts = transcribe_stable(model, wave, _span_bounds=[(int(a), int(b)) for a, b in spans], **kw)
    return merge_span_results(parts, kw.get("language"))
