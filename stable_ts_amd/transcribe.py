"""model.transcribe(): the 30-s window loop around the GPU hot path.

Mirrors ``transcribe_stable`` (whisper_word_level/original_whisper.py:27-781); the recording is pulled window by
window from an ``AudioLoader`` (audio_io.py: waveforms, WAVE paths / bytes, streamed or whole) as the reference does
(:289-311, :494):
per window  log-mel -> encoder -> decode_with_fallback (:349-393) -> segment slicing at consecutive timestamp tokens
(:550-602) -> segment filters (:604-627) -> word timestamps (:635-652) -> instant-word / probability filters (:654-674)
-> seek advance (:629-633, :703-704), with the prompt carried over between windows (:533, :680-682, :706-708).

Two drivers share one per-batch routine:
  * sequential (default): identical control flow to the reference, one window per iteration;
  * ``batch_size=N`` (window-parallel): fixed 30-s stride, no prompt carry-over, N windows per GPU batch -- the mode
    SURVEY.md 8e describes for throughput / sharding; its oracle is "the reference run on each 30-s clip separately".
Out of scope here: yt-dlp URLs, denoisers, VAD models, resume (non-WAVE containers need ffmpeg on PATH).
"""
import time
import warnings
from dataclasses import replace
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .stabilization import host_single_thread
from .audio import CHUNK_LENGTH, HOP_LENGTH, N_FRAMES, N_SAMPLES, N_SAMPLES_PER_TOKEN, SAMPLE_RATE
from .audio_io import AudioLoader, audioloader_not_supported, prep_audio
from .decoding import DecodingOptions, DecodingPlan, DecodingResult
from .result import WhisperResult
from .timing import APPEND_PUNCTUATIONS, PREPEND_PUNCTUATIONS, add_word_timestamps_batch
from .tokenizer import get_tokenizer

# Diagnostic only (bench.py --phase-times): when set to a dict, _process_batch synchronises the device at each stage boundary
# and accumulates wall seconds per stage, so that the host-side share of a pass can be read off.  None in normal operation
# (the product path never synchronises between the stages).
PHASE_TIMES: Optional[dict] = None


def _phase(name: str, t0: float) -> float:
    import time
    if PHASE_TIMES is None:
        return t0
    torch.cuda.synchronize()
    t = time.perf_counter()
    PHASE_TIMES[name] = PHASE_TIMES.get(name, 0.0) + (t - t0)
    return t

_DECODE_KEYS = set(DecodingOptions.__dataclass_fields__)


def as_waveform(audio, only_voice_freq: bool = False) -> torch.Tensor:
    """Whole recording as a 1-D f32 tensor (device preserved for tensors): arrays / tensors are taken as 16 kHz already
    (2-D = channels first, down-mixed), paths and file bytes are decoded by ``audio_io.prep_audio``
    (alignment.py:167-176 and the other whole-file callers of the reference's ``prep_audio``)."""
    audioloader_not_supported(audio)
    if isinstance(audio, np.ndarray):
        audio = torch.from_numpy(np.ascontiguousarray(audio))
    if torch.is_tensor(audio):
        audio = audio.float().mean(0) if audio.ndim == 2 else audio.to(torch.float32)
    return prep_audio(audio, only_voice_freq=only_voice_freq).to(torch.float32)


def pop_audio_options(options: dict) -> dict:
    """Takes the audio pre-processing options of the reference's whole-file entry points (align / align_words / refine
    / locate) out of ``options`` and returns the keyword arguments for :func:`as_waveform`.  Denoisers are out of scope
    (DESIGN.md section 7); ``stream`` has no effect on a recording that is held whole."""
    for k in ("denoiser", "demucs"):
        if options.pop(k, None):
            raise NotImplementedError(f"{k} is outside this package's scope (DESIGN.md section 7)")
    for k in ("denoiser_options", "demucs_options", "stream", "only_ffmpeg"):
        options.pop(k, None)
    return dict(only_voice_freq=bool(options.pop("only_voice_freq", False)))


def _xkv_select(model, xkv, idx: Sequence[int]):
    """Sub-batch of a cross-KV buffer [L][W*1500][2d] (device-side gather; memory movement only)."""
    W = xkv.n_windows
    if list(idx) == list(range(W)):
        return xkv
    d = model.dims
    t = xkv.view(model.engine.tdtype).view(d.n_text_layer, W, -1)      # [L][W][K | V^T] chunks
    sel = t.index_select(1, torch.tensor(list(idx), device=t.device)).contiguous().view(torch.uint8).view(-1)
    sel.n_windows = len(idx)
    return sel


def _decode_with_fallback(model, xkv, base: dict, temperatures: Sequence[float], prompts, ts_masks,
                          compression_ratio_threshold, logprob_threshold, no_speech_threshold,
                          uids: Optional[Sequence[int]] = None, torch_rng: bool = False) -> List[DecodingResult]:
    """original_whisper.py:349-393, for W windows: every window walks the temperature ladder independently; the ones that
    still need a fallback are re-decoded together at the next temperature.  ``torch_rng`` (the sequential driver, one window
    per call): sampled retries draw from torch's generator call for call like the reference's loop (Engine.decode), so with
    the same ``torch.manual_seed`` they are the reference's tokens; otherwise the draws are keyed on ``uids``."""
    W = xkv.n_windows
    results: List[Optional[DecodingResult]] = [None] * W
    pending = list(range(W))
    for t in temperatures:
        kw = dict(base)
        if t > 0:
            kw.pop("beam_size", None)
            kw.pop("patience", None)
        else:
            kw.pop("best_of", None)
        options = DecodingOptions(**kw, temperature=t)
        sub = _xkv_select(model, xkv, pending)
        plans = [DecodingPlan(model, replace(options, prompt=(list(prompts[w]) if prompts[w] else None))) for w in pending]
        groups = {}
        for k, p in enumerate(plans):
            groups.setdefault((p.sample_begin, p.sample_len, p.options.min_tokens), []).append(k)
        outs: List[Optional[DecodingResult]] = [None] * len(pending)
        for _, ks in groups.items():       # one lockstep job per distinct initial length
            sub_k = _xkv_select(model, sub, ks)
            masks = None
            if ts_masks is not None:
                masks = torch.stack([ts_masks[pending[k]] for k in ks])
            # sampling draws are keyed on the window's identity (its seek position), not on its row in this batch
            out = model.engine.decode(sub_k, [list(plans[k].initial_tokens) for k in ks], ts_mask=masks,
                                      window_uid=None if uids is None else [uids[pending[k]] for k in ks],
                                      **(dict(torch_rng=True) if torch_rng and t > 0 and W == 1 else {}),
                                      **plans[ks[0]].engine_kwargs())
            for k, r in zip(ks, plans[ks[0]].results(out, [None] * len(ks), [options.language or "en"] * len(ks))):
                outs[k] = r
        nxt = []
        for k, w in enumerate(pending):
            r = outs[k]
            results[w] = r
            need = False
            if compression_ratio_threshold is not None and r.compression_ratio > compression_ratio_threshold:
                need = True
            if logprob_threshold is not None and r.avg_logprob < logprob_threshold:
                need = True
            if no_speech_threshold is not None and r.no_speech_prob > no_speech_threshold:
                need = False
            if need:
                nxt.append(w)
        pending = nxt
        if not pending:
            break
    return results


def _slice_segments(tokens: List[int], result: DecodingResult, tokenizer, time_offset: float, seek_sample: int,
                    segment_duration: float, time_precision: float):
    """original_whisper.py:406-421, 550-602: cut the window's tokens into segments at consecutive timestamp tokens."""
    tb = tokenizer.timestamp_begin
    tk = np.asarray(tokens, dtype=np.int64)
    is_ts = tk >= tb
    single_ts_ending = is_ts[-2:].tolist() == [False, True]
    consecutive = (np.where(is_ts[:-1] & is_ts[1:])[0] + 1).tolist()

    def seg(start, end, toks):
        toks = [int(t) for t in toks]
        return dict(seek=round(seek_sample / SAMPLE_RATE, 3), start=start, end=end,
                    text=tokenizer.decode([t for t in toks if t < tokenizer.eot]), tokens=toks,
                    temperature=result.temperature, avg_logprob=result.avg_logprob,
                    compression_ratio=result.compression_ratio, no_speech_prob=result.no_speech_prob)

    segs = []
    end_ts_pos = 0
    if consecutive:
        cuts = list(consecutive)
        if single_ts_ending:
            cuts.append(len(tk))
        last = 0
        for cut in cuts:
            sl = tk[last:cut]
            start_pos = int(sl[0]) - tb
            end_ts_pos = int(sl[-1]) - tb
            segs.append(seg(round(time_offset + start_pos * time_precision, 3),
                            round(time_offset + min(end_ts_pos * time_precision, segment_duration), 3), sl))
            last = cut
    else:
        duration = segment_duration
        ts = tk[is_ts]
        if len(ts) > 0 and int(ts[-1]) != tb:
            end_ts_pos = int(ts[-1]) - tb
            duration = min(end_ts_pos * time_precision, segment_duration)
        else:
            end_ts_pos = 0
        segs.append(seg(round(time_offset, 3), round(time_offset + duration, 3), tk))
    return segs, single_ts_ending, end_ts_pos


def _audio_key(a: torch.Tensor):
    return (a.data_ptr(), int(a.shape[-1]))


def _start_encoder(model, audios: List[torch.Tensor]) -> dict:
    """Enqueues spectrogram, encoder and cross-K/V projection of some windows WITHOUT waiting for them (none of the three
    entry points synchronises): the window-parallel driver calls this before the host half of the silence analysis, which
    then runs while the device works.  `_process_batch` picks the results up for the windows that are still in the batch
    unchanged (the usual case); a window the analysis skips or truncates is simply not used / encoded again."""
    seg_samples = [int(a.shape[-1]) for a in audios]
    mel = model.log_mel_batch(audios, [max(N_SAMPLES - n, 0) for n in seg_samples])          # :528-530
    xkv = model.cross_kv(model.encoder(mel))
    return dict(keys=[_audio_key(a) for a in audios], mel=mel, xkv=xkv)


def _process_batch(model, tokenizer, batch: List[dict], o: dict, pre: Optional[dict] = None) -> List[dict]:
    """Runs the hot path for a batch of windows.  batch[w] = dict(audio=1-D f32 tensor (<=480000), seek_sample=int,
    prompt=list[int], ts_mask=bool[1501] | None).  Returns per window dict(segments, segment_samples, result, skipped).
    ``pre``: what `_start_encoder` enqueued earlier for (a superset of) these windows."""
    W = len(batch)
    audios = [b["audio"] for b in batch]
    seg_samples = [int(a.shape[-1]) for a in audios]
    import time
    t_ph = time.perf_counter() if PHASE_TIMES is not None else 0.0
    mel = xkv = None
    if pre is not None:
        where = {k: i for i, k in enumerate(pre["keys"])}
        idx = [where.get(_audio_key(a)) for a in audios]
        if all(i is not None for i in idx):
            same = idx == list(range(len(pre["keys"])))
            mel = pre["mel"] if same else pre["mel"][idx]
            xkv = pre["xkv"] if same else _xkv_select(model, pre["xkv"], idx)
    if xkv is None:
        mel = model.log_mel_batch(audios, [max(N_SAMPLES - n, 0) for n in seg_samples])          # :528-530
        t_ph = _phase("mel", t_ph)
        xkv = model.cross_kv(model.encoder(mel))
    t_ph = _phase("encoder+cross_kv", t_ph)
    ts_masks = [b.get("ts_mask") for b in batch] if o["suppress_ts_tokens"] else None
    if ts_masks is not None and all(m is None for m in ts_masks):
        ts_masks = None
    elif ts_masks is not None:
        ts_masks = [torch.zeros(1501, dtype=torch.bool) if m is None else m for m in ts_masks]
    results = _decode_with_fallback(model, xkv, o["decode_options"], o["temperatures"], [b["prompt"] for b in batch],
                                    ts_masks, o["compression_ratio_threshold"], o["logprob_threshold"],
                                    o["no_speech_threshold"], uids=[int(b["seek_sample"]) // 160 for b in batch],
                                    torch_rng=bool(o.get("torch_sampling")))
    t_ph = _phase("decode (device loop + result copy)", t_ph)
    time_precision = (N_FRAMES // model.dims.n_audio_ctx) * HOP_LENGTH / SAMPLE_RATE
    punct = o["prepend_punctuations"] + o["append_punctuations"]
    outs = []
    for w in range(W):
        r = results[w]
        out = dict(segments=[], segment_samples=seg_samples[w], result=r, skipped=False, single_ts_ending=False,
                   num_samples=seg_samples[w])
        outs.append(out)
        if o["no_speech_threshold"] is not None:                                                # :537-546
            skip = r.no_speech_prob > o["no_speech_threshold"]
            if o["logprob_threshold"] is not None and r.avg_logprob > o["logprob_threshold"]:
                skip = False
            if skip:
                out["skipped"] = True
                continue
        if len(r.tokens) == 0:
            out["skipped"] = True
            continue
        time_offset = batch[w]["seek_sample"] / SAMPLE_RATE
        seg_dur = seg_samples[w] / SAMPLE_RATE
        segs, single_end, end_ts_pos = _slice_segments(r.tokens, r, tokenizer, time_offset, batch[w]["seek_sample"],
                                                       seg_dur, time_precision)
        for i in reversed(range(len(segs))):                                                    # :604-627
            s = segs[i]
            if s["text"].strip() in punct:
                del segs[i]
            elif o["word_timestamps"]:
                if s["start"] == s["end"]:
                    del segs[i]
            else:
                nxt = i + 1
                max_end = s["end"] if nxt >= len(segs) else segs[nxt]["start"]
                if s["start"] > s["end"]:
                    if i != 0 and segs[i - 1]["end"] != segs[i - 1]["start"] and segs[i - 1]["end"] < max_end:
                        s["start"] = segs[i - 1]["end"]
                    else:
                        s["start"] = max_end
        out["segments"] = segs
        out["single_ts_ending"] = single_end
        out["num_samples"] = (min(round(end_ts_pos * N_SAMPLES_PER_TOKEN), seg_samples[w]) if end_ts_pos > 0
                              else seg_samples[w])                                              # :629-633

    t_ph = _phase("host: segment slicing", t_ph)
    if o["word_timestamps"]:
        idx = [w for w in range(W) if outs[w]["segments"]]
        if idx:
            add_word_timestamps_batch(                                                          # :635-652
                model=model, tokenizer=tokenizer,
                windows=[dict(segments=outs[w]["segments"], num_samples=outs[w]["num_samples"]) for w in idx],
                xkv=_xkv_select(model, xkv, idx), prepend_punctuations=o["prepend_punctuations"],
                append_punctuations=o["append_punctuations"],     # min_word_dur stays at the callee's 0.1 (:636-652)
                split_callback=o["split_callback"], gap_padding=o["gap_padding"], dynamic_heads=o.get("dynamic_heads"),
                aligner=o.get("aligner", "legacy"), extra_models=o.get("extra_models"),
                mel=mel[idx] if o.get("extra_models") else None)
        t_ph = _phase("word timestamps (scoring pass + a7 + DTW + host word assembly)", t_ph)
        for w in idx:
            out = outs[w]
            segs = out["segments"]
            for i in reversed(range(len(segs))):                                                # :654-663
                words = segs[i]["words"]
                zero = np.array([wd["start"] == wd["end"] for wd in words]).astype(np.float16).mean()
                if zero > o["max_instant_words"]:
                    del segs[i]
            if o["avg_prob_threshold"] and segs:                                                # :665-674
                time_offset = batch[w]["seek_sample"] / SAMPLE_RATE
                if out["single_ts_ending"] and (np.mean([wd["probability"] for s in segs for wd in s["words"]])
                                                < o["avg_prob_threshold"]):
                    out["num_samples"] = seg_samples[w]
                    segs.clear()
                else:
                    out["num_samples"] = round((segs[-1]["words"][-1]["end"] - time_offset) * SAMPLE_RATE)
    for w in range(W):
        out = outs[w]
        if not out["segments"]:
            continue
        if not out["single_ts_ending"] or o["avg_prob_threshold"]:                              # :703-704
            out["segment_samples"] = out["num_samples"]
    return outs


@host_single_thread
def transcribe_stable(model, audio, *, verbose: Optional[bool] = False,
                      temperature: Union[float, Tuple[float, ...]] = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0),
                      compression_ratio_threshold: Optional[float] = 2.4, logprob_threshold: Optional[float] = -1.0,
                      no_speech_threshold: Optional[float] = 0.6, condition_on_previous_text: bool = True,
                      initial_prompt: Optional[str] = None, word_timestamps: bool = True,
                      regroup: Union[bool, str] = True, suppress_silence: bool = True, suppress_word_ts: bool = True,
                      use_word_position: bool = True, q_levels: int = 20, k_size: int = 5, vad: bool = False,
                      min_word_dur: Optional[float] = 0.1, min_silence_dur: Optional[float] = None,
                      nonspeech_error: float = 0.1, prepend_punctuations: Optional[str] = None,
                      append_punctuations: Optional[str] = None, suppress_ts_tokens: bool = False,
                      gap_padding: str = " ...", max_instant_words: float = 0.5, avg_prob_threshold: Optional[float] = None,
                      nonspeech_skip: Optional[float] = None, progress_callback: Callable = None,
                      ignore_compatibility: bool = True, split_callback: Callable = None,
                      batch_size: Optional[int] = None, clip_timestamps: Optional[Union[str, List[float]]] = None,
                      streams: int = 1, stream: Optional[bool] = None, only_voice_freq: bool = False,
                      only_ffmpeg: bool = False, denoiser: Optional[str] = None, denoiser_options: Optional[dict] = None,
                      extra_models: Optional[list] = None, dynamic_heads: Optional[Union[bool, int, str]] = None,
                      aligner: Union[str, dict] = "legacy", _span_bounds: Optional[List[Tuple[int, int]]] = None,
                      **decode_options) -> WhisperResult:
    """Same keyword surface as the reference's ``model.transcribe`` for the options that reach the hot path
    (original_whisper.py:27-79); ``batch_size`` (window-parallel mode) and ``streams`` are the only additions.
    ``audio``: waveform (tensor / array, 16 kHz), file path or file bytes, or an ``AudioLoader``; ``stream`` loads files
    in chunks as the seek advances (default for paths, as in the reference)."""
    unknown = set(decode_options) - _DECODE_KEYS
    if unknown:
        raise TypeError(f"transcribe() got unexpected keyword argument(s): {sorted(unknown)}")
    if vad:
        raise NotImplementedError("vad=True needs the Silero model (torch.hub, network) -- out of scope offline")
    decode_options = dict(decode_options)
    decode_options["fp16"] = model.engine.dtype_name == "f16"
    if "max_initial_timestamp" not in decode_options:
        decode_options["max_initial_timestamp"] = None                                         # :262-263
    task = decode_options.get("task", "transcribe")

    # clip_timestamps (original_whisper.py:280-287): pairs of seconds; only those sections are loaded and processed
    if isinstance(clip_timestamps, str):
        clip_timestamps = [float(ts) for ts in (clip_timestamps.split(",") if clip_timestamps else [])]
    sections = None
    if clip_timestamps:
        if batch_size:
            raise NotImplementedError("clip_timestamps is defined on the sequential driver (window-parallel mode: slice the audio)")
        sections = [list(clip_timestamps[i:i + 2]) for i in range(0, len(clip_timestamps), 2)]
        if len(sections[-1]) == 1:
            sections[-1] = [sections[-1][0], None]
    if isinstance(audio, AudioLoader):                                                          # :289-298
        audio.validate_external_args(sr=SAMPLE_RATE, vad=vad, stream=stream, denoiser=denoiser,
                                     denoiser_options=denoiser_options, only_voice_freq=only_voice_freq)
        audio.load_sections = sections
        loader = audio
    else:                                                                                       # :299-311
        if torch.is_tensor(audio) or isinstance(audio, np.ndarray):
            audio = as_waveform(audio)
        loader = AudioLoader(audio, stream=stream, denoiser=denoiser, denoiser_options=denoiser_options,
                             only_voice_freq=only_voice_freq, only_ffmpeg=only_ffmpeg, verbose=verbose,
                             new_chunk_divisor=None, load_sections=sections)

    # Language and tokenizer are settled at the FIRST WINDOW THAT IS ACTUALLY DECODED (original_whisper.py:319-345, called at
    # :532 after the silent windows were skipped, after nonspeech_skip trimmed the window and inside the first clip section) --
    # not on the first 30 s of the file.  `settle_language` is called with that window's audio right before its batch runs.
    lang_state = dict(language=decode_options.get("language"), tokenizer=None, prompt_tokens=[])
    if lang_state["language"] or not model.is_multilingual:
        lang_state["language"] = lang_state["language"] or "en"
        decode_options["language"] = lang_state["language"]
        lang_state["tokenizer"] = get_tokenizer(model.is_multilingual, num_languages=model.num_languages,
                                                language=lang_state["language"], task=task)
        if initial_prompt is not None:                                                          # :342-345
            lang_state["prompt_tokens"] = lang_state["tokenizer"].encode(" " + initial_prompt.strip())
    initial_prompt_tokens: List[int] = lang_state["prompt_tokens"]

    def settle_language(first_audio: torch.Tensor, tracks_: list):
        if lang_state["tokenizer"] is not None:
            return
        n = int(first_audio.shape[-1])
        mel0 = model.log_mel(first_audio, max(N_SAMPLES - n, 0))
        _, probs = model.detect_language(mel0)
        lang = max(probs, key=probs.get)
        lang_state["language"] = decode_options["language"] = lang
        tok = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=lang, task=task)
        lang_state["tokenizer"] = tok
        if initial_prompt is not None:
            lang_state["prompt_tokens"].extend(tok.encode(" " + initial_prompt.strip()))
            for tr in tracks_:                      # the reference extends all_tokens at this point (:344-345)
                tr.all_tokens.extend(lang_state["prompt_tokens"])

    def new_track(source: AudioLoader, offset: int = 0) -> _Track:
        from .stabilization import NonSpeechPredictor
        # :427-441: one predictor per run, always (an all-zero window is fast-forwarded in every mode); the
        # loudness-based detector only with suppress_silence, otherwise it looks at exact-zero samples and yields no
        # timings (so nonspeech_skip has nothing to act on, as in the reference)
        predictor = NonSpeechPredictor(q_levels=q_levels, k_size=k_size, min_word_dur=min_word_dur,
                                       min_silence_dur=min_silence_dur, get_mask=suppress_ts_tokens,
                                       loudness=bool(suppress_silence))
        return _Track(source, predictor, list(initial_prompt_tokens), offset)

    # one track = one run of the reference's sequential algorithm; ``_span_bounds`` (transcribe_spans) makes several
    if _span_bounds:
        whole = loader.next_chunk(0, loader.get_total_samples())
        tracks = [new_track(AudioLoader(whole[a0:b0], new_chunk_divisor=None), a0) for a0, b0 in _span_bounds]
    else:
        tracks = [new_track(loader)]
    tr0 = tracks[0]
    nonspeech = tr0.nonspeech

    o = dict(decode_options=decode_options,
             temperatures=[temperature] if isinstance(temperature, (int, float)) else list(temperature),
             compression_ratio_threshold=compression_ratio_threshold, logprob_threshold=logprob_threshold,
             no_speech_threshold=no_speech_threshold, word_timestamps=word_timestamps,
             prepend_punctuations=PREPEND_PUNCTUATIONS if prepend_punctuations is None else prepend_punctuations,
             append_punctuations=APPEND_PUNCTUATIONS if append_punctuations is None else append_punctuations,
             min_word_dur=0.1 if min_word_dur is None else min_word_dur, split_callback=split_callback,
             gap_padding=gap_padding, max_instant_words=max_instant_words, avg_prob_threshold=avg_prob_threshold,
             suppress_ts_tokens=suppress_ts_tokens, extra_models=extra_models, dynamic_heads=dynamic_heads, aligner=aligner,
             # the reference's own control flow (one track, one window per decode call) samples from torch's generator like the
             # reference; window-parallel / span modes decode several windows per call and key the draws on the window instead
             torch_sampling=not batch_size and not _span_bounds)

    def host_copy(seg: torch.Tensor) -> torch.Tensor:
        return seg.detach().float().cpu()                       # silence analysis is host-side vector code (CPU)

    def predict_nonspeech(tr_ns, chunks: List[torch.Tensor], offsets: List[float], pool_=None, between=None) -> List[dict]:
        """The silence analysis of some windows.  Windows that are resident on the GPU go through the device probe (k-th
        largest level + the ~6000 samples the loudness curve reads; engine.loudness_probe) and only the curve's arithmetic
        runs on the host -- the same expressions on the same values, so the masks are those of the full-length host path,
        which is what windows on the host, the exact-zero mode (suppress_silence=False) and probe refusals still take."""
        from .stabilization import loudness_from_probe
        probes = [False] * len(chunks)
        if tr_ns.loudness and chunks and all(c.is_cuda for c in chunks):
            from .engine import loudness_probe
            probes = loudness_probe(chunks)
        if between is not None:
            between()                      # device work that may run under the host half below

        def one(a):
            ch, off, pr = a
            if pr is False:
                return tr_ns.predict(host_copy(ch), offset=off)
            loud = None if pr is None else loudness_from_probe(*pr)
            if loud is False:
                return tr_ns.predict(host_copy(ch), offset=off)
            return tr_ns.predict(None, offset=off, loud=loud)

        args = list(zip(chunks, offsets, probes))
        # the probe leaves ~0.8 ms of small tensor operations per window: a thread pool only adds contention there (measured:
        # 20 windows 15 ms in line, 35-48 ms through 8 threads); the full-length path (2-5 ms per window in numpy / torch
        # calls that release the GIL) is the one that gains from the pool
        use_pool = pool_ is not None and len(args) > 1 and any(pr is False for pr in probes)
        return list(pool_.map(one, args)) if use_pool else [one(a) for a in args]

    def window_input(tr: _Track, seek: int, seg: torch.Tensor, prompt: List[int], pred: Optional[dict] = None):
        item = dict(audio=seg, seek_sample=seek, prompt=prompt, ts_mask=None, silence=None, skip=False)
        if tr.nonspeech is not None:
            if pred is None:
                pred = predict_nonspeech(tr.nonspeech, [seg], [seek / SAMPLE_RATE])[0]
            item["silence"] = pred["timings"] if suppress_silence else None
            item["ts_mask"] = pred["mask"]
            item["skip"] = pred["is_silent"]
            if nonspeech_skip and pred["timings"] is not None and not item["skip"]:               # :513-526
                starts = pred["timings"][0] - seek / SAMPLE_RATE
                ends = pred["timings"][1] - seek / SAMPLE_RATE
                long_idx = np.flatnonzero((ends - starts) >= nonspeech_skip)
                if len(long_idx):
                    k = long_idx[0]
                    if starts[k] < o["min_word_dur"] or int(starts[k] * SAMPLE_RATE) == 0:
                        item["skip"] = True
                        item["skip_samples"] = round(ends[k] * SAMPLE_RATE)
                    else:
                        item["audio"] = seg[: int(starts[k] * SAMPLE_RATE)]
        return item

    def commit(tr: _Track, item: dict, out: dict):
        """original_whisper.py:676-708 for one finished window; returns the samples to advance by."""
        segs = out["segments"]
        if not segs:
            return out["segment_samples"]
        tr.all_tokens.extend(t for s in segs for t in s["tokens"])
        if item["silence"] is not None:
            from .stabilization import suppress_segment_silence
            for s in segs:
                suppress_segment_silence(s, *item["silence"], min_word_dur=o["min_word_dur"], word_level=suppress_word_ts,
                                         nonspeech_error=nonspeech_error, use_word_position=use_word_position)
        for s in segs:
            tr.all_segments.append({"id": len(tr.all_segments), **s})
        return out["segment_samples"]

    def next_live_item(tr: _Track) -> Optional[dict]:
        """the head of one iteration of the reference's loop (:494-526): fetch the window at the track's seek, fast-forward
        over windows the silence analysis skips; None when the track has no audio left"""
        while True:
            seg, tr.seek = tr.loader.next_valid_chunk(tr.seek, N_SAMPLES)                        # :494-500
            if seg is None:
                return None
            item = window_input(tr, tr.seek, seg, tr.all_tokens[tr.prompt_reset_since:])
            n_seg = int(item["audio"].shape[-1])
            if n_seg == 0:
                return None
            if not item["skip"]:
                tr.started = True
                return item
            tr.seek += item.get("skip_samples", n_seg)

    def advance(tr: _Track, item: dict, out: dict):
        adv = commit(tr, item, out)
        if out["segments"]:
            if not condition_on_previous_text or out["result"].temperature > 0.5:                # :706-708
                tr.prompt_reset_since = len(tr.all_tokens)
        n_seg = int(item["audio"].shape[-1])
        if adv is not None and int(adv) <= 0:
            # the reference adds 0 to its seek here and never returns (:629-633 with avg_prob_threshold when the last
            # word ends at the window start); a recording must not be able to hang the loop: skip the window instead
            warnings.warn(f"window at {tr.seek / SAMPLE_RATE:.2f}s produced no forward progress; skipping it")
            adv = n_seg
        tr.seek += int(adv) if adv is not None else n_seg

    # Ctrl-C ends the run with what has been transcribed so far (original_whisper.py:712-723): the result's
    # ``unfinished_start`` is the time up to which it is complete (-1 = finished)
    interrupted: List[float] = []

    def _interrupted_time(tr: _Track, seek_sample: int) -> float:
        t = tr.all_segments[-1]["end"] if tr.all_segments else -1.0
        return max(t, seek_sample / SAMPLE_RATE)

    if batch_size:
        # ---- window-parallel driver: fixed stride, no prompt carry-over
        # the loader hands out the windows in order (a streamed source only moves forward); chunks are short-lived
        def batches():
            pos, group = 0, []
            while (chunk := loader.next_chunk(pos, N_SAMPLES)) is not None:
                group.append((pos, chunk))
                pos += N_SAMPLES
                if len(group) == batch_size:
                    yield group
                    group = []
            if group:
                yield group

        pool = None
        if nonspeech is not None:
            # the windows of a batch are known up front in this mode: analyse them concurrently (torch/numpy release the GIL)
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=8)
        lanes = None
        if streams and streams > 1:
            # experimental: each batch is split over `streams` host threads, each driving its own HIP stream through its
            # own engine clone (shared weights).  The decode step is a chain of short latency-bound kernels; two
            # independent chains can overlap on the device.  Results are identical (windows are independent in this mode).
            from concurrent.futures import ThreadPoolExecutor
            lanes = [model] + [model.clone_for_stream() for _ in range(streams - 1)]
            lane_pool = ThreadPoolExecutor(max_workers=streams)

        def run_lane(m, part):
            if not part:
                return []
            ctx, side = m.stream_context()
            with ctx:
                outs_k = _process_batch(m, lang_state["tokenizer"], part, o)
            if side is not None:
                side.synchronize()
            return outs_k

        done = 0
        try:
            for group in batches():
                preds = [None] * len(group)
                t_ph = time.perf_counter() if PHASE_TIMES is not None else 0.0
                pre = {}
                if pool is not None and len(group) > 1:
                    def start_device_work():
                        force = getattr(model, "prestart_encoder", None)     # tests switch it on for the host stand-in
                        if not lanes and (force or (force is None and all(g[1].is_cuda for g in group))):
                            pre.update(_start_encoder(model, [g[1] for g in group]))
                    preds = predict_nonspeech(nonspeech, [g[1] for g in group], [g[0] / SAMPLE_RATE for g in group], pool,
                                              between=start_device_work)
                items = [window_input(tr0, sk, ch, list(initial_prompt_tokens), pr) for (sk, ch), pr in zip(group, preds)]
                t_ph = _phase("host: silence analysis of the batch (copy-out + loudness, 8 threads)", t_ph)
                live = [it for it in items if not it["skip"] and it["audio"].shape[-1] > 0]
                tr0.started = tr0.started or bool(live)
                if live:
                    settle_language(live[0]["audio"], [tr0])
                    for it in live:
                        if not it["prompt"]:
                            it["prompt"] = list(lang_state["prompt_tokens"])
                tokenizer = lang_state["tokenizer"]
                if lanes and len(live) >= len(lanes):
                    per = (len(live) + len(lanes) - 1) // len(lanes)
                    parts = [live[k * per:(k + 1) * per] for k in range(len(lanes))]
                    outs = [x for r in lane_pool.map(run_lane, lanes, parts) for x in r]
                else:
                    outs = _process_batch(model, tokenizer, live, o, pre=pre or None) if live else []
                t_ph = time.perf_counter() if PHASE_TIMES is not None else 0.0
                for it, out in zip(live, outs):
                    commit(tr0, it, out)
                t_ph = _phase("host: commit (silence suppression of the word times)", t_ph)
                done += len(items)
                if progress_callback is not None:
                    total = loader.get_total_samples()
                    progress_callback(min(total, done * N_SAMPLES) / SAMPLE_RATE, total / SAMPLE_RATE)
        except KeyboardInterrupt:                                                               # :716-723
            interrupted.append(_interrupted_time(tr0, done * N_SAMPLES))
        if lanes:
            lane_pool.shutdown()
        if pool is not None:
            pool.shutdown()
    else:
        # ---- sequential driver (reference control flow); with several tracks the tracks advance in lockstep, one window
        # of each per device batch -- every track still sees exactly the reference's sequence of windows and prompts
        active = list(tracks)
        while active:
            try:
                items = [(tr, it) for tr in active if (it := next_live_item(tr)) is not None]
                if not items:
                    break
                if lang_state["tokenizer"] is None:
                    settle_language(items[0][1]["audio"], tracks)
                    for tr, it in items:                # the prompt slices were taken before the initial prompt was known
                        it["prompt"] = tr.all_tokens[tr.prompt_reset_since:]
                outs = _process_batch(model, lang_state["tokenizer"], [it for _, it in items], o)
            except KeyboardInterrupt:                                                           # :716-723
                interrupted.append(_interrupted_time(tracks[0], tracks[0].seek))
                break
            for (tr, it), out in zip(items, outs):
                advance(tr, it, out)
            active = [tr for tr, _ in items]
            if progress_callback is not None:
                total = loader.get_total_samples()
                at = sum(min(tr.seek, tr.loader.get_total_samples()) for tr in tracks)
                progress_callback(min(at, total) / SAMPLE_RATE, total / SAMPLE_RATE)
    loader.terminate()                                                                          # :731

    def finish(tr: _Track) -> WhisperResult:
        language = lang_state["language"]
        tokenizer = lang_state["tokenizer"]
        if tokenizer is None:       # nothing was ever decoded (all windows silent): no language was settled (:527)
            tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language="en", task=task)
        text = tokenizer.decode(tr.all_tokens[len(initial_prompt_tokens):])
        # the reference settles the language at the first window that is not skipped as silent (:527, detect_language());
        # a recording without such a window yields language None
        result = WhisperResult(dict(text=text, segments=tr.all_segments, language=language if tr.started else None,
                                    time_scale=None), force_order=not word_timestamps)                    # :735-743
        timings = tr.nonspeech.timings() if (tr.nonspeech is not None and suppress_silence) else None
        if timings:
            result.update_nonspeech_sections(*timings, overwrite=True)                            # :770-771
        if word_timestamps and regroup:
            from .regroup import regroup_default
            regroup_default(result, regroup)
        return result

    if _span_bounds:
        return [(tr.offset, finish(tr)) for tr in tracks]
    t_ph = time.perf_counter() if PHASE_TIMES is not None else 0.0
    result = finish(tr0)
    _phase("host: result assembly (WhisperResult, non-speech sections)", t_ph)
    if interrupted:
        result.unfinished_start = interrupted[0]                                                # :776
    if len(result.text) == 0:
        warnings.warn(f"Failed to {task} audio. Result contains no text. ")
    return result


@host_single_thread
def transcribe_minimal(model, audio, *, verbose: Optional[bool] = False, word_timestamps: bool = True,
                       regroup: Union[bool, str] = True, suppress_silence: bool = True, suppress_word_ts: bool = True,
                       use_word_position: bool = True, q_levels: int = 20, k_size: int = 5, denoiser: Optional[str] = None,
                       denoiser_options: Optional[dict] = None, demucs: bool = False, demucs_options: Optional[dict] = None,
                       vad: bool = False, vad_threshold: float = 0.35, vad_onnx: bool = False, min_word_dur: float = 0.1,
                       nonspeech_error: float = 0.1, only_voice_freq: bool = False, only_ffmpeg: bool = False,
                       **options) -> WhisperResult:
    """``model.transcribe_minimal`` (original_whisper.py:784-928): the plain recogniser wrapped by ``transcribe_any``'s pre- and
    post-processing (voice-frequency filter before, silence adjustment and regrouping after) instead of the stabilising
    window loop.  The reference's inner recogniser is upstream ``whisper.transcribe``; here it is this package's window loop
    with every stabilising option switched off (no silence analysis inside the loop, no timestamp-token suppression, no
    regrouping) -- the same device hot path per window.  Options that ``transcribe_any`` understands go to it, the rest to
    the recogniser, as in the reference (``isolate_useful_options``)."""
    import inspect
    from .audio_io import audioloader_not_supported
    from .non_whisper import transcribe_any
    audioloader_not_supported(audio)
    any_keys = set(inspect.signature(transcribe_any).parameters)
    extra = {k: options.pop(k) for k in list(options) if k in any_keys}
    if not isinstance(audio, (str, bytes)):
        extra.setdefault("input_sr", SAMPLE_RATE)
    if denoiser or only_voice_freq:
        extra.setdefault("audio_type", "torch")
        extra.setdefault("model_sr", SAMPLE_RATE)

    def recognise(audio, **kw):
        return transcribe_stable(model, audio, verbose=verbose, word_timestamps=word_timestamps, regroup=False,
                                 suppress_silence=False, suppress_ts_tokens=False, **kw)

    return transcribe_any(inference_func=recognise, audio=audio, inference_kwargs=dict(options), verbose=verbose,
                          regroup=regroup, suppress_silence=suppress_silence, suppress_word_ts=suppress_word_ts,
                          q_levels=q_levels, k_size=k_size, denoiser=denoiser, denoiser_options=denoiser_options,
                          demucs=demucs, demucs_options=demucs_options, vad=vad, vad_threshold=vad_threshold,
                          vad_onnx=vad_onnx, min_word_dur=min_word_dur, nonspeech_error=nonspeech_error,
                          use_word_position=use_word_position, only_voice_freq=only_voice_freq, only_ffmpeg=only_ffmpeg,
                          force_order=True, **extra)


class _Track:
    """State of one run of the reference's sequential window loop (seek, prompt history, segments, silence analysis)."""
    __slots__ = ("loader", "nonspeech", "all_tokens", "all_segments", "prompt_reset_since", "seek", "offset", "started")

    def __init__(self, loader: AudioLoader, nonspeech, prompt_tokens: List[int], offset: int = 0):
        self.loader, self.nonspeech, self.all_tokens, self.offset = loader, nonspeech, prompt_tokens, offset
        self.all_segments: List[dict] = []
        self.prompt_reset_since = 0
        self.seek = 0
        self.started = False
