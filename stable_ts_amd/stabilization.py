"""Non-VAD silence detection and timestamp snapping (default-on in the reference: ``suppress_silence=True``).

Host-side vector code on <=480000 samples per window; not part of the GPU hot path but it moves word boundaries by up
to hundreds of ms, so it is restated for drop-in behaviour.  Follows
stabilization/nonvad.py:16-88 (loudness quantisation), stabilization/utils.py:43-111 (mask <-> timings),
stabilization/__init__.py:16-135,241-254 (NonSpeechPredictor, non-VAD branch) and :300-379 (suppress_silence),
result.py:681-705 (per-word ``keep_end`` policy).  Silero VAD (``vad=True``) needs torch.hub + network: out of scope.
"""
import functools
import threading
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from ._num import round3
from .audio import FRAMES_PER_SECOND, N_SAMPLES_PER_TOKEN, TOKENS_PER_SECOND
from .timing import APPEND_PUNCTUATIONS


class _single_host_thread:
    """The full-length host path below runs a handful of element-wise tensor operations over 480 000 samples; with torch's
    default intra-op pool every one of them wakes all OpenMP workers, which then busy-wait for more work for a while.  In a
    container with a CPU quota (the MI355X boxes this was measured on) those spinning workers get the whole process
    throttled 45-65 ms at a time -- `align()` of a host waveform ran at 580x real time with them and 1 270x without
    (DESIGN.md section 5).  The operations are far too small to gain from threads, so they run on the calling thread."""

    # torch.set_num_threads is process-wide for the intra-op pool: nested / concurrent uses are counted under a lock and only
    # the outermost entry saves and the last exit restores the caller's setting
    _lock = threading.Lock()
    _depth = 0
    _saved = 1

    def __enter__(self):
        cls = _single_host_thread
        with cls._lock:
            if cls._depth == 0:
                cls._saved = torch.get_num_threads()
                if cls._saved != 1:
                    torch.set_num_threads(1)
            cls._depth += 1

    def __exit__(self, *exc):
        cls = _single_host_thread
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0 and cls._saved != 1:
                torch.set_num_threads(cls._saved)
        return False


def host_single_thread(fn):
    """Decorator of the entry points whose arithmetic is all on the GPU (model.transcribe / align / align_words / refine / locate /
    transcribe_spans): everything this package computes on the host there is small (token bookkeeping, 1501-point loudness
    curves, index lists).  torch's intra-op pool is parked for the duration of the call (and restored after), for the reason
    given at `_single_host_thread`.  NOT used on `transcribe_any`: its `inference_func`, denoisers and callbacks are the
    caller's code and may well be CPU torch models -- there only this package's own silence analysis parks the pool."""
    @functools.wraps(fn)
    def wrapped(*args, **kw):
        # a model object that does its arithmetic on the host (the CPU stand-in of the test-suite: tests/oracle_engine.py)
        # says so and keeps the pool -- there the pool IS the compute
        if args and getattr(args[0], "computes_on_host", False):
            return fn(*args, **kw)
        with _single_host_thread():
            return fn(*args, **kw)
    return wrapped


def audio2loudness(x: torch.Tensor) -> Optional[torch.Tensor]:
    """nonvad.py:16-39: |x| normalised by (1.75 x the 99.9th-percentile level), resampled to one value per 20 ms."""
    x = x.abs()
    k = int(x.numel() * 0.001)
    if k:
        # the k-th largest value (== torch.topk(x, k).values[-1], the reference's expression) by O(n) selection
        xn = x.numpy()
        thr = torch.tensor(np.partition(xn, xn.size - k)[xn.size - k])
    else:
        thr = x.quantile(0.999, dim=-1)
    units = round(x.shape[-1] / N_SAMPLES_PER_TOKEN) + 1
    if units <= 2:
        return None
    if thr < 1e-5:
        return torch.zeros(units, dtype=x.dtype, device=x.device)
    x = x / min(1.0, float(thr) * 1.75)
    return F.interpolate(x[None, None], size=units, mode="linear", align_corners=False)[0, 0]


def mask2timing(mask, time_offset: float = 0.0) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """utils.py:43-86 (no clipping arguments): runs of True -> (starts, ends) in seconds at 50 units/s."""
    if mask is None or len(mask) == 0 or not bool(mask.any()):
        return None
    m = mask.cpu().numpy().copy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
    p = np.concatenate(([False], m, [False]))
    starts = np.logical_and(~p[:-2], p[1:-1]).nonzero()[0] / TOKENS_PER_SECOND
    ends = (np.logical_and(p[1:-1], ~p[2:]).nonzero()[0] + 1) / TOKENS_PER_SECOND
    if time_offset:
        starts = starts + time_offset
        ends = ends + time_offset
    return starts, ends


def timing2mask(starts: np.ndarray, ends: np.ndarray, size: int) -> torch.Tensor:
    """utils.py:89-111."""
    out = torch.zeros(size, dtype=torch.bool)
    a = (starts * TOKENS_PER_SECOND).round().astype(np.int32)
    b = (ends * TOKENS_PER_SECOND).round().astype(np.int32)
    for i, j in zip(a, b):
        out[i:j + 1] = True
    return out


@functools.lru_cache(maxsize=64)
def probe_indices(n: int) -> Optional[np.ndarray]:
    """The sample indices that ``F.interpolate(|x|, size=units, mode='linear')`` of ``audio2loudness`` reads for a window of
    ``n`` samples (two per 20-ms unit: ``floor(src)`` and its right neighbour, ``src = scale (i + 0.5) - 0.5`` in float32),
    plus one more sample on either side so that a last-bit difference in ``src`` cannot matter.  int32 [4 * units], or None
    when the window is too short for a mask (nonvad.py:27-29)."""
    units = round(n / N_SAMPLES_PER_TOKEN) + 1
    if units <= 2:
        return None
    scale = np.float32(n) / np.float32(units)
    src = np.maximum(scale * (np.arange(units, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5), np.float32(0.0))
    i0 = src.astype(np.int64)
    idx = np.stack([i0 - 1, i0, i0 + 1, i0 + 2], axis=1).clip(0, n - 1)
    return np.ascontiguousarray(idx.reshape(-1).astype(np.int32))


_probe_scratch = threading.local()


def loudness_from_probe(n: int, thr: float, idx: np.ndarray, vals: torch.Tensor) -> Union[torch.Tensor, None, bool]:
    """``audio2loudness`` of a window of ``n`` samples from the device probe (``swx_loudness_probe``): ``thr`` = the k-th
    largest |x| (bit for bit the value the host selection returns), ``vals`` = |x| at ``idx = probe_indices(n)``.  The
    division and the interpolation run through the SAME torch expressions as the full-length path, on a length-``n`` array
    that holds the gathered samples at their positions and NaN everywhere else: the interpolation reads two samples per
    output, so the result is the full path's bit for bit -- and if it ever read a sample that was not gathered the NaN
    would show, in which case False is returned and the caller takes the full path."""
    units = round(n / N_SAMPLES_PER_TOKEN) + 1
    if units <= 2:
        return None
    if not np.isfinite(thr):
        return False                                   # k == 0 (fewer than 1000 samples): the quantile branch, host path
    thr_t = torch.tensor(np.float32(thr))
    if thr_t < 1e-5:
        return torch.zeros(units, dtype=torch.float32)
    cache = getattr(_probe_scratch, "buf", None)
    if cache is None:
        cache = _probe_scratch.buf = {}
    x = cache.get(n)
    if x is None:
        if len(cache) > 4:
            cache.clear()
        x = torch.full((n,), float("nan"), dtype=torch.float32)
        if n <= 2 * 480000:                    # only window-sized scratch is kept (<= 3.8 MB each, five per thread)
            cache[n] = x
    x[torch.from_numpy(idx.astype(np.int64))] = vals / min(1.0, float(thr_t) * 1.75)
    out = F.interpolate(x[None, None], size=units, mode="linear", align_corners=False)[0, 0]
    if bool(torch.isnan(out).any()):
        return False
    return out


def wav2mask(audio: Optional[torch.Tensor], q_levels: int = 20, k_size: int = 5, *, loud=False) -> Optional[torch.Tensor]:
    """nonvad.py:43-88: boolean SILENCE mask per 20-ms unit, or None when the window has no silence.  ``loud``: the window's
    loudness curve when it was computed elsewhere (``loudness_from_probe``); False = compute it from ``audio``."""
    if loud is False:
        loud = audio2loudness(audio)
    if loud is None:
        return None
    p = k_size // 2 if k_size else 0
    if p and p < loud.shape[-1]:
        m = torch.avg_pool1d(F.pad(loud[None], (p, p), "reflect"), kernel_size=k_size, stride=1)[0]
    else:
        m = loud.clone()
    if q_levels:
        m = m.mul(q_levels).round()
    speech = m.bool()
    if not speech.any():
        return ~speech
    s, e = mask2timing(speech)
    keep = (e - s) > 0.1                    # speech runs of <= 0.1 s are treated as silence
    silence = ~timing2mask(s[keep], e[keep], loud.shape[-1])
    if not silence.any():
        return None
    return silence


class NonSpeechPredictor:
    """Non-VAD branches of stabilization/__init__.py::NonSpeechPredictor (:16-135, 241-286).

    ``loudness=True``  -> ``predict_with_nonvad`` (the default ``suppress_silence=True`` path): quantised loudness mask.
    ``loudness=False`` -> ``predict_with_samples`` (``suppress_silence=False``): no timings, the mask only looks at
    exact-zero samples.  ``pad_mask`` pads the mask to the 1501 timestamp tokens of a window (transcribe's
    ``mask_pad_func``); the aligner uses the mask unpadded."""

    def __init__(self, q_levels: int = 20, k_size: int = 5, min_word_dur: Optional[float] = 0.1,
                 min_silence_dur: Optional[float] = None, get_mask: bool = False, pad_mask: bool = True,
                 loudness: bool = True):
        self.q_levels, self.k_size = q_levels, k_size
        self.min_silence_dur = min_silence_dur
        self.get_mask = get_mask
        self.pad_mask = pad_mask
        self.loudness = loudness
        mwd = 0.1 if min_word_dur is None else min_word_dur
        self.min_units_per_word = max(round(mwd * FRAMES_PER_SECOND), 1)
        self.min_samples_per_word = round(mwd * 16000)
        self._starts: List[float] = []
        self._ends: List[float] = []

    def _pad(self, mask: torch.Tensor) -> torch.Tensor:
        if not self.pad_mask:
            return mask
        pad = torch.zeros(1501, dtype=torch.bool)
        n = min(1501, mask.shape[-1])
        pad[:n] = mask[:n]
        return pad

    def predict(self, audio: Optional[torch.Tensor], offset: float = 0.0, *, loud=False) -> dict:
        """``loud``: the loudness curve of the window from the device probe (``loudness_from_probe``; None = window too short
        for a mask); with it ``audio`` is not read."""
        if loud is False and self.loudness and audio is not None and audio.is_cuda and audio.numel() <= 2 * 480000:
            # (windows only: a whole recording handed in on the device takes the full-length path below -- the probe's NaN-filled
            # scratch of the window length is cached per thread, which is fine for 1.9 MB and not for an hour of samples)
            # a window that is resident on the GPU: k-th largest level + the samples the curve reads come from the device
            # probe (24 KB instead of a 1.9 MB copy-out and a host selection); same values, same arithmetic
            from .engine import loudness_probe
            pr = loudness_probe([audio.reshape(-1)])[0]
            got = None if pr is None else loudness_from_probe(*pr)
            if got is not False:
                loud = got
        if loud is not False and self.loudness:
            return self._from_mask(wav2mask(None, self.q_levels, self.k_size, loud=loud), offset)
        # (the public entry points run with torch's intra-op pool parked -- host_single_thread -- so the element-wise passes
        # over 480 000 samples below do not wake OpenMP workers; a direct caller decides for itself)
        audio = audio.detach().float().cpu().contiguous()
        if not self.loudness:
            # :271-286 with get_mask: one flag per 20-ms unit, True where EVERY sample of the unit is non-zero
            if not self.get_mask:
                # :277-281: without a mask the window counts as silent when fewer than min_word_dur worth of samples
                # are non-zero
                silent = bool(int(audio.count_nonzero()) < self.min_samples_per_word)
                return dict(timings=None, mask=None, is_silent=silent)
            extra = audio.shape[-1] % N_SAMPLES_PER_TOKEN
            if extra:
                audio = F.pad(audio, (0, N_SAMPLES_PER_TOKEN - extra))
            mask = torch.all(audio.reshape(-1, N_SAMPLES_PER_TOKEN) != 0, dim=-1)
            silent = bool((mask.shape[-1] - int(mask.count_nonzero())) < self.min_units_per_word)
            return dict(timings=None, mask=self._pad(mask), is_silent=silent)
        return self._from_mask(wav2mask(audio, self.q_levels, self.k_size), offset)

    def _from_mask(self, mask: Optional[torch.Tensor], offset: float) -> dict:
        timings = mask2timing(mask, time_offset=offset)
        if timings is not None:
            timings = np.stack(timings, axis=0)
        is_silent = False
        if mask is not None:
            is_silent = bool((mask.shape[-1] - int(mask.count_nonzero())) < self.min_units_per_word)
            mask = self._pad(mask)
        if timings is not None and len(timings[0]):
            self._starts.extend(timings[0].tolist())
            self._ends.extend(timings[1].tolist())
        if self.min_silence_dur and timings is not None:
            keep = (timings[1] - timings[0]) >= self.min_silence_dur
            timings = np.stack((timings[0][keep], timings[1][keep]), axis=0)
        return dict(timings=timings, mask=mask if self.get_mask else None, is_silent=is_silent)

    def timings(self) -> Optional[Tuple[List[float], List[float]]]:
        """finalize_timings (:120-135): every section seen so far, sorted, overlaps merged; None when none was seen."""
        if not self._starts:
            return None
        s, e = np.sort(np.array(self._starts)), np.sort(np.array(self._ends))
        while len(s) > 1:
            ok = s[1:] >= e[:-1]
            if ok.all():
                break
            s = s[np.concatenate(([True], ok))]
            e = e[np.concatenate((ok, [True]))]
        return s.tolist(), e.tolist()

    def sections(self) -> List[dict]:
        t = self.timings()
        return [] if t is None else [dict(start=float(a), end=float(b)) for a, b in zip(*t)]


class _DictSpan:
    """start/end attribute view of a dict (the snapping code below is written against attributes)."""
    __slots__ = ("d",)

    def __init__(self, d: dict):
        self.d = d

    start = property(lambda self: self.d["start"], lambda self, v: self.d.__setitem__("start", v))
    end = property(lambda self: self.d["end"], lambda self, v: self.d.__setitem__("end", v))


def snap_to_speech(obj, starts: np.ndarray, ends: np.ndarray, min_word_dur: float, nonspeech_error: float,
                   keep_end: Optional[bool]):
    """stabilization/__init__.py:300-379 on any object with ``start`` / ``end`` attributes: a start lying in a
    non-speech section moves to the section's end, an end lying in one moves to its start (never below
    ``min_word_dur``); a single section strictly inside the span is cut off from the side it nearly touches."""
    starts, ends = np.asarray(starts), np.asarray(ends)
    if len(starts) == 0 or (obj.end - obj.start) <= min_word_dur:
        return
    # every rule below needs a section that reaches into the span (ends > start and starts < end, or it touches an edge):
    # most words of a transcript have none, and this one test spares them the three masked searches
    if not np.any((ends >= obj.start) & (starts <= obj.end)):
        return
    if keep_end is None or keep_end:
        hit = np.all((starts <= obj.start, obj.start < ends, ends <= obj.end), axis=0).nonzero()[0]
        if len(hit):
            obj.start = min(float(ends[hit[0]]), round3(obj.end - min_word_dur))
            if (obj.end - obj.start) <= min_word_dur:
                return
    if not keep_end:
        hit = np.all((obj.start <= starts, starts < obj.end, obj.end <= ends), axis=0).nonzero()[0]
        if len(hit):
            obj.end = max(float(starts[hit[0]]), round3(obj.start + min_word_dur))
            if (obj.end - obj.start) <= min_word_dur:
                return
    if nonspeech_error:
        inside = np.logical_and(obj.start <= starts, obj.end >= ends).nonzero()[0]
        if len(inside) != 1:
            return
        s0, e0 = float(starts[inside[0]]), float(ends[inside[0]])
        dur = e0 - s0
        err_start = (s0 - obj.start) / dur
        err_end = (obj.end - e0) / dur
        ke = keep_end if keep_end is not None else (err_start <= err_end)
        if not (err_start <= nonspeech_error or err_end <= nonspeech_error):
            return
        if ke:
            obj.start = min(e0, round3(obj.end - min_word_dur))
        else:
            obj.end = max(s0, round3(obj.start + min_word_dur))


def _snap(obj: dict, starts, ends, min_word_dur, nonspeech_error, keep_end):
    snap_to_speech(_DictSpan(obj), starts, ends, min_word_dur, nonspeech_error, keep_end)


def suppress_segment_silence(seg: dict, starts, ends, min_word_dur: float = 0.1, word_level: bool = True,
                             nonspeech_error: float = 0.1, use_word_position: bool = True):
    """result.py:681-705 on a segment dict (words = list of dicts); keeps segment start/end in sync with the words."""
    starts, ends = np.asarray(starts), np.asarray(ends)
    words = seg.get("words")
    if words:
        sel = words if word_level or len(words) == 1 else [words[0], words[-1]]
        for i, w in enumerate(sel, 1):
            keep_end = (not (w["word"][-1] in APPEND_PUNCTUATIONS or i == len(sel))) if use_word_position else None
            _snap(w, starts, ends, min_word_dur, nonspeech_error, keep_end)
        for w in words:                                      # WordTiming stores millisecond-rounded stamps (result.py:38-41)
            w["start"], w["end"] = (round3(w["start"]) if w["start"] else w["start"]), (round3(w["end"]) if w["end"] else w["end"])
        seg["start"], seg["end"] = words[0]["start"], words[-1]["end"]
        # the reference round-trips the dict through Segment(...).to_dict() here (original_whisper.py:684-695): with
        # words present, text and tokens become views of the words (timestamp tokens drop out of ``tokens``)
        seg["text"] = "".join(w["word"] for w in words)
        if words[0].get("tokens"):
            seg["tokens"] = [t for w in words for t in w["tokens"]]
    else:
        _snap(seg, starts, ends, min_word_dur, nonspeech_error, True)
        seg["start"], seg["end"] = (round3(seg["start"]) if seg["start"] else 0.0), (round3(seg["end"]) if seg["end"] else 0.0)
