"""model.align() / model.align_words(): forced alignment of given text (stable_whisper/alignment.py:27-368).

Per window the device work is mel -> encoder -> one teacher-forced decoder pass with the alignment heads' qk captured
-> softmax / z-norm / median / head-mean -> DTW; no autoregressive decoding.  The boundary the
reference defines for this is ``compute_timestamps(audio_segment, word_tokens)`` (alignment.py:405-429, seam B2):
`make_alignment_func` returns exactly that callable, and `align` drives it with `stable_ts_amd.aligner.Aligner`, the
restatement of the reference's window state machine (non_whisper/alignment.py:58-1033: token batching by
``token_step``, gap padding, non-speech skipping, the re-alignment policy of ``_fallback``), which is tested against
the reference's own class on CPU (tests/test_aligner_cpu.py).
"""
from typing import List, Optional, Sequence, Union

import torch

from .stabilization import host_single_thread
from .audio import N_SAMPLES, SAMPLE_RATE
from .result import WhisperResult
from .timing import add_word_timestamps_batch
from .tokenizer import get_tokenizer


from .aligner import WordToken  # noqa: E402,F401  (re-exported: the seam-B2 callable takes these)


def make_alignment_func(model, tokenizer, extra_models: Optional[list] = None, dynamic_heads=None,
                        aligner: Union[str, dict] = "legacy"):
    """alignment.py:396-429: inference_func(audio_segment f32[n<=480000], word_tokens) -> list of word dicts with times
    relative to the segment start.  ``extra_models`` / ``dynamic_heads`` / ``aligner`` select the head-selection variants
    of the attention stage (timing.py)."""

    def compute_timestamps(audio_segment: torch.Tensor, word_tokens: List[WordToken]) -> List[dict]:
        return compute_timestamps_batch([audio_segment], [word_tokens])[0]

    def compute_timestamps_batch(audio_segments: Sequence[torch.Tensor], word_tokens_list: Sequence[List[WordToken]]):
        from . import transcribe as _tr                     # diagnostic stage timer (bench.py --phase-times), off by default
        import time
        t_ph = time.perf_counter() if _tr.PHASE_TIMES is not None else 0.0
        n = [int(a.shape[-1]) for a in audio_segments]
        mel = model.log_mel_batch(list(audio_segments), [max(N_SAMPLES - k, 0) for k in n])
        t_ph = _tr._phase("align: audio upload + mel", t_ph)
        xa = model.encoder(mel)
        t_ph = _tr._phase("align: encoder", t_ph)
        xkv = model.cross_kv(xa)
        t_ph = _tr._phase("align: cross K/V", t_ph)
        windows = []
        for k, wts in zip(n, word_tokens_list):
            seg = dict(seek=0, tokens=([w.word for w in wts], [list(w.tokens) for w in wts]))
            windows.append(dict(segments=[seg], num_samples=k))
        add_word_timestamps_batch(model=model, tokenizer=tokenizer, windows=windows, xkv=xkv,
                                  split_callback=lambda x, _: x, gap_padding=None,
                                  prepend_punctuations="", append_punctuations="", extra_models=extra_models,
                                  dynamic_heads=dynamic_heads, aligner=aligner, mel=mel if extra_models else None)
        _tr._phase("align: word timestamps (scoring pass + a7 + DTW + host)", t_ph)
        return [w["segments"][0]["words"] for w in windows]

    compute_timestamps.batch = compute_timestamps_batch
    return compute_timestamps


def make_refinement_func(model, tokenizer):
    """alignment.py:636-672 (seam B3): ``inference_func(audio_segment f32[2, n], tokens) -> probabilities
    [2, len(tokens), eot]`` over the text vocabulary, a device tensor.  Both audio copies go through mel -> encoder ->
    cross-KV and ONE teacher-forced decoder pass of ``sot_sequence + [no_timestamps] + tokens + [eot]``.
    Like the reference the log-mel is computed on the un-padded segment and the remaining frames are filled with 0.0
    (``pad_or_trim`` of the mel, not of the audio).  Host-side bisection: stable_ts_amd/refiner.py (CPU-tested against
    the reference's Refiner); the callable is CPU-tested against the reference's on the engine stand-in
    (tests/test_locate_cpu.py) and has a hardware check (tests/hw_checks/b3_check.py; status in DESIGN.md section 7)."""
    sot = list(tokenizer.sot_sequence)

    def inference_func(audio_segment: torch.Tensor, tokens: List[int]) -> torch.Tensor:
        audio_segment = audio_segment[..., :N_SAMPLES]      # the Refiner's probes are word-local, far below 30 s
        mel = model.log_mel_segments([audio_segment[0], audio_segment[1]], batch_max=True)
        xkv = model.cross_kv(model.encoder(mel))
        ids = [*sot, tokenizer.no_timestamps, *[int(t) for t in tokens], tokenizer.eot]
        logits = model.engine.forward_logits(xkv, [ids, ids])
        return logits[:, len(sot): len(sot) + len(tokens), : tokenizer.eot].softmax(dim=-1)

    return inference_func


@host_single_thread
def align(model, audio, text: Union[str, List[int], WhisperResult], language: str = None, *, tokenizer=None,
          ignore_compatibility: bool = False, remove_instant_words: bool = False, token_step: int = 100,
          original_split: bool = False, word_dur_factor: Optional[float] = 2.0, max_word_dur: Optional[float] = 3.0,
          nonspeech_skip: Optional[float] = 5.0, fast_mode: bool = False, failure_threshold: Optional[float] = None,
          batch_size: int = 1, **options) -> Optional[WhisperResult]:
    """Forced alignment of ``text`` (plain text, token ids, or a WhisperResult whose text is re-aligned) with ``audio``
    (alignment.py:27-216).  The window state machine -- token batching, gap padding, non-speech skipping, the
    re-alignment policy -- is :class:`stable_ts_amd.aligner.Aligner`, a restatement of the reference's ``Aligner`` that is
    checked against it window for window on CPU; each window's timestamps come from the device through
    ``make_alignment_func`` (seam B2).  Returns None when nothing could be aligned."""
    from .aligner import Aligner
    from .transcribe import as_waveform, pop_audio_options
    audio_options = pop_audio_options(options)
    max_step = model.dims.n_text_ctx - 6                       # alignment.py:181-185
    if token_step < 1:
        token_step = max_step
    elif token_step > max_step:
        raise ValueError(f"The max value for [token_step] is {max_step} but got {token_step}.")
    if tokenizer is None:
        if not language and model.is_multilingual and (language := getattr(text, "language", None)) is None:
            raise TypeError("expected argument for language")                                  # alignment.py:375-383
        tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=language or "en",
                                  task="transcribe")
    lang_code = getattr(tokenizer, "language_code", None) or getattr(tokenizer, "language", None)
    variant = {k: options.pop(k) for k in ("extra_models", "dynamic_heads", "aligner") if k in options}
    aligner = Aligner(inference_func=make_alignment_func(model, tokenizer, **variant), decode=tokenizer.decode, encode=tokenizer.encode,
                      split_words_by_space=lang_code not in {"zh", "ja", "th", "lo", "my"}, sample_rate=SAMPLE_RATE,
                      max_segment_length=N_SAMPLES, remove_instant_words=remove_instant_words, token_step=token_step,
                      original_split=original_split, word_dur_factor=word_dur_factor, max_word_dur=max_word_dur,
                      nonspeech_skip=nonspeech_skip, fast_mode=fast_mode, failure_threshold=failure_threshold, **options)
    # the waveform stays where it is: windows of a recording that is resident on the GPU are analysed there (device probe of
    # the silence analysis) and are not uploaded again
    result = aligner.align(as_waveform(audio, **audio_options).detach().float(), text)
    if result is not None:
        result.language = lang_code or language or (None if model.is_multilingual else "en")    # alignment.py:388-393
    return result


@host_single_thread
def align_words(model, audio, result: Union[WhisperResult, List[dict]], language: str = None, *,
                ignore_compatibility: bool = False, tokenizer=None, normalize_text: bool = True, inplace: bool = True,
                batch_size: int = 8, **options) -> WhisperResult:
    """alignment.py:219-368 / non_whisper/alignment.py:396-474: (re-)time the words of every segment inside the
    segment's own start/end.  Segments are independent, so ``batch_size`` of them share one encoder / scoring pass on
    the device (the reference runs them one by one); the host logic is ``Aligner.align_words``, compared with the
    reference's on CPU, batched and unbatched."""
    from .aligner import Aligner
    from .transcribe import as_waveform, pop_audio_options
    # the fallback machinery of align() does not exist here: its options are refused like any unknown keyword
    # (options.py:16-18 via AllOptions at alignment.py:350)
    extras = [k for k in options if k in ("remove_instant_words", "token_step", "original_split", "word_dur_factor",
                                          "max_word_dur", "nonspeech_skip", "fast_mode", "failure_threshold")]
    if extras:
        raise TypeError(f"got unexpected keyword argument(s): {', '.join(extras)}")
    audio_options = pop_audio_options(options)
    if tokenizer is None:
        language = language or getattr(result, "language", None)
        if not language and model.is_multilingual:
            raise TypeError("expected argument for language")
        tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=language or "en",
                                  task="transcribe")
    lang_code = getattr(tokenizer, "language_code", None) or getattr(tokenizer, "language", None)
    variant = {k: options.pop(k) for k in ("extra_models", "dynamic_heads", "aligner") if k in options}
    func = make_alignment_func(model, tokenizer, **variant)

    def clipped(fn):        # a segment longer than one window is aligned against its first 30 s (the reference trims the mel)
        return lambda chunks, words: fn([c[..., :N_SAMPLES] for c in chunks], words)

    aligner = Aligner(inference_func=lambda seg, words: clipped(func.batch)([seg], [words])[0], decode=tokenizer.decode,
                      encode=tokenizer.encode, split_words_by_space=lang_code not in {"zh", "ja", "th", "lo", "my"},
                      sample_rate=SAMPLE_RATE, max_segment_length=N_SAMPLES, token_step=model.dims.n_text_ctx, **options)
    out = aligner.align_words(as_waveform(audio, **audio_options).detach().float(), result, normalize_text, inplace,
                              batch_inference=clipped(func.batch), batch_size=batch_size)
    out.language = lang_code or language or (None if model.is_multilingual else "en")
    return out

@host_single_thread
def refine(model, audio, result: WhisperResult, *, steps: str = None, rel_prob_decrease: float = .03,
           abs_prob_decrease: float = .05, rel_rel_prob_decrease: Optional[float] = None, prob_threshold: float = .5,
           rel_dur_change: Optional[float] = .5, abs_dur_change: Optional[float] = None, word_level: bool = True,
           precision: float = None, single_batch: bool = False, inplace: bool = True, **options) -> WhisperResult:
    """alignment.py:512-635: move word starts later / ends earlier as far as the token probabilities allow.  The
    bisection is :class:`stable_ts_amd.refiner.Refiner`; ``single_batch`` is accepted for signature compatibility (the
    two audio copies always share one batched pass here)."""
    from .refiner import Refiner
    from .transcribe import as_waveform, pop_audio_options
    audio_options = pop_audio_options(options)
    if result and (not result.has_words or any(w.probability is None for w in result.all_words())):
        if not result.language:
            raise RuntimeError("cannot align words with result missing language")
        result = align_words(model, audio, result)
    tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=result.language or "en",
                              task="transcribe")
    if result and not all(w.tokens for w in result.all_words()):
        for w in result.all_words():
            w.tokens = tokenizer.encode(w.word)
    refiner = Refiner(make_refinement_func(model, tokenizer), sample_rate=SAMPLE_RATE, steps=steps,
                      rel_prob_decrease=rel_prob_decrease, abs_prob_decrease=abs_prob_decrease,
                      rel_rel_prob_decrease=rel_rel_prob_decrease, prob_threshold=prob_threshold,
                      rel_dur_change=rel_dur_change, abs_dur_change=abs_dur_change, word_level=word_level,
                      precision=precision, max_inference_tokens=model.dims.n_text_ctx - 6, **options)
    return refiner.refine(as_waveform(audio, **audio_options), result, inplace)
