"""model.align() / model.align_words(): forced alignment of given text (stable_whisper/alignment.py:27-368).

Per window the device work is mel -> encoder -> one teacher-forced decoder pass with the alignment heads' qk captured
-> softmax / z-norm / median / head-mean -> DTW; no autoregressive decoding.  The boundary the
reference defines for this is ``compute_timestamps(audio_segment, word_tokens)`` (alignment.py:405-429, seam B2):
`make_alignment_func` returns exactly that callable, and `align` drives it with a compact restatement of the
window loop of non_whisper/alignment.py:252-394 (token batching by ``token_step``, seek = end of the last word whose
timing is trusted).  The reference's re-alignment heuristics (`_fallback`, :937-1006) and non-speech skipping
(:873-935) are "next" items.
"""
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from .audio import N_SAMPLES, SAMPLE_RATE
from .result import WhisperResult
from .timing import add_word_timestamps_batch
from .tokenizer import get_tokenizer


class WordToken:
    def __init__(self, word: str, tokens: List[int], is_padding: bool = False):
        self.word, self.tokens, self.is_padding = word, tokens, is_padding


def make_alignment_func(model, tokenizer):
    """alignment.py:396-429: inference_func(audio_segment f32[n<=480000], word_tokens) -> list of word dicts with times
    relative to the segment start."""

    def compute_timestamps(audio_segment: torch.Tensor, word_tokens: List[WordToken]) -> List[dict]:
        return compute_timestamps_batch([audio_segment], [word_tokens])[0]

    def compute_timestamps_batch(audio_segments: Sequence[torch.Tensor], word_tokens_list: Sequence[List[WordToken]]):
        n = [int(a.shape[-1]) for a in audio_segments]
        mel = model.log_mel_batch(list(audio_segments), [max(N_SAMPLES - k, 0) for k in n])
        xkv = model.cross_kv(model.encoder(mel))
        windows = []
        for k, wts in zip(n, word_tokens_list):
            seg = dict(seek=0, tokens=([w.word for w in wts], [list(w.tokens) for w in wts]))
            windows.append(dict(segments=[seg], num_samples=k))
        add_word_timestamps_batch(model=model, tokenizer=tokenizer, windows=windows, xkv=xkv,
                                  split_callback=lambda x, _: x, gap_padding=None,
                                  prepend_punctuations="", append_punctuations="")
        return [w["segments"][0]["words"] for w in windows]

    compute_timestamps.batch = compute_timestamps_batch
    return compute_timestamps


def _words_from_text(text: str, tokenizer) -> List[WordToken]:
    tokens = tokenizer.encode(text if text.startswith(" ") else " " + text.strip())
    words, groups = tokenizer.split_to_word_tokens(tokens)
    return [WordToken(w, g) for w, g in zip(words, groups) if len(g)]


def align(model, audio, text: Union[str, List[int], WhisperResult], language: str = None, *, token_step: int = 100,
          tokenizer=None, batch_size: int = 1, regroup: Union[bool, str] = True,
          **options) -> Optional[WhisperResult]:
    """Forced alignment.  ``text`` may be a string, a token list, or a WhisperResult (its text is re-aligned).
    Windows are consumed sequentially: each call aligns up to ``token_step`` tokens against the next <=30 s of audio and
    the seek moves to the end of the last word that ended before the window's final second."""
    from .transcribe import load_audio
    max_step = model.dims.n_text_ctx - 6                       # alignment.py:181-185
    if token_step < 1:
        token_step = max_step
    elif token_step > max_step:
        raise ValueError(f"The max value for [token_step] is {max_step} but got {token_step}.")
    if tokenizer is None:
        if language is None and model.is_multilingual and not isinstance(text, WhisperResult):
            raise TypeError("expected argument for language")
        if isinstance(text, WhisperResult) and language is None:
            language = text.language
        tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=language or "en",
                                  task="transcribe")
    if isinstance(text, WhisperResult):
        text = text.text
    if isinstance(text, str):
        queue = _words_from_text(text, tokenizer)
    else:
        words, groups = tokenizer.split_to_word_tokens(list(text))
        queue = [WordToken(w, g) for w, g in zip(words, groups)]
    audio = load_audio(audio)
    total = int(audio.shape[-1])
    func = make_alignment_func(model, tokenizer)
    done: List[dict] = []
    seek = 0
    while queue and seek < total:
        seg = audio[seek: seek + N_SAMPLES]
        take, n_tok = [], 0
        for w in queue:
            if take and n_tok + len(w.tokens) > token_step:
                break
            take.append(w)
            n_tok += len(w.tokens)
        timed = func(seg, take)
        seg_dur = seg.shape[-1] / SAMPLE_RATE
        offset = seek / SAMPLE_RATE
        last_window = seek + N_SAMPLES >= total
        # trust words that end before the last second of the window (the tail is re-aligned with more context)
        n_keep = len(timed)
        if not last_window:
            n_keep = 0
            for wd in timed:
                if wd["end"] <= seg_dur - 1.0:
                    n_keep += 1
                else:
                    break
            n_keep = max(n_keep, 1)
        for wd in timed[:n_keep]:
            done.append(dict(word=wd["word"], start=round(wd["start"] + offset, 3), end=round(wd["end"] + offset, 3),
                             probability=wd["probability"], tokens=wd["tokens"]))
        queue = queue[n_keep:]
        new_seek = int(round(done[-1]["end"] * SAMPLE_RATE))
        seek = new_seek if new_seek > seek else seek + int(seg.shape[-1])
    if not done:
        return None
    for w in queue:                                           # unaligned tail: zero-length words at EOF (:349-362)
        t = round(total / SAMPLE_RATE, 3)
        done.append(dict(word=w.word, start=t, end=t, probability=0.0, tokens=w.tokens))
    seg = dict(start=done[0]["start"], end=done[-1]["end"], text="".join(w["word"] for w in done), seek=0.0,
               tokens=[t for w in done for t in w["tokens"]], words=done)
    result = WhisperResult(dict(segments=[seg], language=getattr(tokenizer, "language", language)), check_sorted=False)
    if regroup:                                               # non_whisper/alignment.py:388-389
        result.regroup(regroup)
    return result


def align_words(model, audio, result: Union[WhisperResult, List[dict]], language: str = None, *, tokenizer=None,
                batch_size: int = 8, regroup: Union[bool, str] = True, **options) -> WhisperResult:
    """alignment.py:219-368: re-align the words of each pre-timed segment independently (embarrassingly parallel:
    segments are batched `batch_size` at a time through one encoder / scoring pass)."""
    from .transcribe import load_audio
    if tokenizer is None:
        lang = language or (result.language if isinstance(result, WhisperResult) else None) or "en"
        tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=lang, task="transcribe")
    segs = [s.to_dict() for s in result.segments] if isinstance(result, WhisperResult) else [dict(s) for s in result]
    audio = load_audio(audio)
    func = make_alignment_func(model, tokenizer)
    jobs = []
    for s in segs:
        a = int(round(s["start"] * SAMPLE_RATE))
        b = min(int(round(s["end"] * SAMPLE_RATE)), a + N_SAMPLES, int(audio.shape[-1]))
        if s.get("words"):
            wts = [WordToken(w["word"], list(w["tokens"]) if w.get("tokens") else tokenizer.encode(w["word"])) for w in s["words"]]
        else:
            wts = _words_from_text(s["text"], tokenizer)
        jobs.append((s, a, b, wts))
    out_segments = []
    for k in range(0, len(jobs), batch_size):
        chunk = [j for j in jobs[k: k + batch_size] if j[2] > j[1] and j[3]]
        if not chunk:
            continue
        timed = func.batch([audio[a:b] for _, a, b, _ in chunk], [w for *_, w in chunk])
        for (s, a, b, _), words in zip(chunk, timed):
            off = a / SAMPLE_RATE
            ws = [dict(word=w["word"], start=round(w["start"] + off, 3), end=round(w["end"] + off, 3),
                       probability=w["probability"], tokens=w["tokens"]) for w in words]
            out_segments.append(dict(start=ws[0]["start"], end=ws[-1]["end"], text="".join(w["word"] for w in ws),
                                     seek=round(off, 3), tokens=[t for w in ws for t in w["tokens"]], words=ws))
    out = WhisperResult(dict(segments=out_segments, language=getattr(tokenizer, "language", language)), check_sorted=False)
    if regroup:                                               # non_whisper/alignment.py:472
        out.regroup(regroup)
    return out
