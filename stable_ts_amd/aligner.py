"""The forced-alignment window state machine behind ``align()`` (row a11 of SURVEY.md section 8).

Behavioural contract = ``stable_whisper/non_whisper/alignment.py::Aligner`` (:58-1033): text -> word tokens
(``tokens_to_word_tokens`` :1065-1095, ``merge_punctuations`` :1035-1062), per window a batch of up to ``token_step``
tokens with gap-padding pseudo-words at sentence ends (``_get_curr_words`` :731-750, ``pad_segment_word_tokens``
:1098-1127), the inference call and its word/text consistency check (``_compute_timestamps`` :657-729), non-speech
skipping (``_skip_nonspeech`` :873-935) and the re-alignment policy that decides which words of a window are trusted and
where the next window starts (``_fallback`` :937-1006, ``_redo_words`` :825-871, ``_fix_temp_words`` :752-788,
``_is_new_better`` :803-816).  It is generic over ``inference_func`` (seam B2): the GPU path plugs in
``stable_ts_amd.alignment.make_alignment_func`` (mel -> encoder -> scoring pass -> DTW on the device), the CPU test
(tests/test_aligner_cpu.py) plugs the SAME synthetic inference function into this class and into the reference's
``Aligner`` and requires identical results, window for window.

Host-only control flow on a few hundred words; there is nothing here for the GPU.  State of a run: ``queue`` = words
still to align, ``pending`` = the last trusted word of the previous window that the next window is allowed to re-time,
``window`` = the words sent to the device for the current window, ``curr`` = their timings.
"""
import re
import warnings
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ._num import round3
from .result import WhisperResult, WordTiming
from .stabilization import NonSpeechPredictor
from .timing import APPEND_PUNCTUATIONS, PREPEND_PUNCTUATIONS


@dataclass
class WordToken:
    word: str
    tokens: List[int]
    is_padding: bool = False


@dataclass
class TimedWord:
    word: str
    start: float
    end: float
    tokens: List[int]
    probability: float


# ------------------------------------------------------------------------------------------------ text -> words
def merge_punctuations(words: List[WordToken], prepend: str = PREPEND_PUNCTUATIONS, append: str = APPEND_PUNCTUATIONS):
    """In place, right to left: a free-standing opening mark (' (') joins the word after it, a closing mark ('.')
    without a leading space joins the word before it (:1035-1062).  Padding pseudo-words are left alone."""
    i = len(words) - 1
    while i >= 0 and len(words) >= 2:
        w = words[i]
        if not w.is_padding:
            if i != len(words) - 1 and w.word.startswith(" ") and w.word.strip() in prepend:
                words.pop(i)
                nxt = words[i]
                nxt.word, nxt.tokens = w.word + nxt.word, w.tokens + nxt.tokens
            w = words[i]
            if i != 0 and not w.word.endswith(" ") and w.word in append:
                words.pop(i)
                prv = words[i - 1]
                prv.word, prv.tokens = prv.word + w.word, prv.tokens + w.tokens
        i -= 1


def tokens_to_word_tokens(tokens: Sequence[int], decode: Callable, split_by_space: bool) -> List[WordToken]:
    """Group tokens into words by incremental decoding (:1065-1095): a token run becomes a unit once its decoding is a
    prefix of what is left of the full text (multi-token code points), a unit without a leading space extends the
    previous word (languages written with spaces), then punctuation is merged."""
    rest: str = decode(list(tokens))
    out: List[WordToken] = []
    run: List[int] = []
    for t in tokens:
        run.append(t)
        piece = decode(run)
        if rest[:len(piece)] != piece:
            continue
        if split_by_space and not piece.startswith(" ") and out:
            out[-1].word += piece
            out[-1].tokens += run
        else:
            out.append(WordToken(piece, run))
        rest = rest[len(piece):]
        run = []
    if run:
        out.append(WordToken(rest, run))
    elif rest:
        out[-1].word += rest
    merge_punctuations(out)
    return out


def standardize_text(text, original_split: bool = False) -> Tuple[Union[str, List[int]], List[int]]:
    """:508-532: whitespace -> single spaces with a leading space; with ``original_split`` the caller's line / segment
    breaks are remembered as cumulative character counts."""
    cuts: List[int] = []
    if isinstance(text, WhisperResult):
        if original_split and len(text.segments) > 1 and text.has_words:
            cuts = np.cumsum([sum(len(w.word) for w in s.words) for s in text.segments]).tolist()
        return text.text, cuts
    if isinstance(text, str):
        if original_split and "\n" in text:
            lines = [" " + n for line in text.splitlines() if (n := re.sub(r"\s", " ", line).strip())]
            return "".join(lines), np.cumsum([len(x) for x in lines]).tolist()
        text = re.sub(r"\s", " ", text)
        return (text if text.startswith(" ") else " " + text), cuts
    return text, cuts


class _MemoryAudio:
    """What the aligner needs from the reference's AudioLoader for an in-memory waveform (audio/__init__.py:301-333,
    402-412): chunks by absolute sample offset, total length, duration."""

    def __init__(self, audio: torch.Tensor, sample_rate: int):
        self.audio, self.sr = audio, sample_rate
        self.total = int(audio.shape[-1])

    def next_chunk(self, seek: int, size: int) -> Optional[torch.Tensor]:
        chunk = self.audio[seek: seek + size]
        return chunk if chunk.shape[-1] else None

    def get_duration(self, ndigits: Optional[int] = None) -> float:
        d = self.total / self.sr
        return d if ndigits is None else round(d, ndigits)

    def get_total_samples(self) -> int:
        return self.total


class Aligner:
    """Same constructor vocabulary as the reference class (:60-250); only the options the window loop reads are kept.
    ``inference_func(audio_segment, word_tokens) -> list of dict(word, start, end, probability, tokens)`` with times
    relative to the segment start; ``decode`` / ``encode`` are the tokenizer's."""

    def __init__(self, inference_func: Callable, decode: Callable, encode: Callable, split_words_by_space: bool = True,
                 sample_rate: int = 16000, max_segment_length: int = 480000, time_precision: float = 0.02,
                 remove_instant_words: bool = False, token_step: int = 100, original_split: bool = False,
                 max_word_dur: Optional[float] = 3.0, word_dur_factor: Optional[float] = 2.0,
                 nonspeech_skip: Optional[float] = 5.0, fast_mode: bool = False,
                 failure_threshold: Optional[float] = None, *, regroup: Union[bool, str] = True,
                 suppress_silence: bool = True, suppress_word_ts: bool = True, use_word_position: bool = True,
                 q_levels: int = 20, k_size: int = 5, min_word_dur: Optional[float] = None,
                 min_silence_dur: Optional[float] = None, nonspeech_error: float = 0.1,
                 presplit: Union[bool, List[str]] = True, gap_padding: Optional[str] = " ...",
                 progress_callback: Optional[Callable] = None, **unsupported):
        if failure_threshold is not None and not 0 <= failure_threshold <= 1:
            raise ValueError(f"``failure_threshold`` ({failure_threshold}) must be between 0 and 1.")
        for k in ("vad", "denoiser", "only_voice_freq", "stream"):
            if unsupported.pop(k, None):
                raise NotImplementedError(f"{k} is outside this package's scope (DESIGN.md section 7)")
        for k in ("verbose", "vad_threshold", "denoiser_options", "prepend_punctuations", "append_punctuations",
                  "dynamic_heads", "aligner", "extra_models", "tokens_per_sec", "all_options"):
            unsupported.pop(k, None)
        if unsupported:
            raise TypeError(f"unexpected keyword argument(s): {', '.join(unsupported)}")      # options.py:16-18
        self.inference_func, self.decode, self.encode = inference_func, decode, encode
        self.split_words_by_space = split_words_by_space
        self.sample_rate, self.n_samples = sample_rate, int(max_segment_length)
        self.tokens_per_sec = round(1 / time_precision)
        self.remove_instant_words = remove_instant_words
        self.token_step, self.original_split = token_step, original_split
        self.max_word_dur, self.word_dur_factor = max_word_dur, word_dur_factor
        self.nonspeech_skip, self.fast_mode, self.failure_threshold = nonspeech_skip, fast_mode, failure_threshold
        self.regroup, self.suppress_silence = regroup, suppress_silence
        self.suppress_word_ts, self.use_word_position = suppress_word_ts, use_word_position
        self.q_levels, self.k_size = q_levels, k_size
        self.min_word_dur = 0.1 if min_word_dur is None else min_word_dur              # default.py:7
        self.min_silence_dur, self.nonspeech_error = min_silence_dur, nonspeech_error
        self.presplit, self.gap_padding = presplit, gap_padding
        self.progress_callback = progress_callback
        self.all_punctuations = PREPEND_PUNCTUATIONS + APPEND_PUNCTUATIONS

    # ------------------------------------------------------------------------------------------------ set-up
    def _load_text(self, text):
        self._text, self._char_cuts = standardize_text(text, self.original_split)
        tokens = self.encode(self._text) if isinstance(self._text, str) else list(self._text)
        self.queue: List[WordToken] = tokens_to_word_tokens(tokens, self.decode, self.split_words_by_space)
        self.total_words = len(self.queue)
        self.chars_left = sum(len(w.word) for w in self.queue)
        self.ends_sentence = self._sentence_end_mask()
        self.failure_count = 0
        self.max_fail = self.total_words * (self.failure_threshold or 1)

    def _sentence_end_mask(self) -> Optional[List[bool]]:
        """One flag per CHARACTER of the text: does the word it belongs to close a sentence (:572-594)?  Indexed from the
        end of the text by the number of characters still queued, which survives re-queueing of words."""
        if not self.presplit:
            return None
        marks = APPEND_PUNCTUATIONS if isinstance(self.presplit, bool) else self.presplit
        mask: List[bool] = []
        if self._char_cuts:
            cuts, seen = list(self._char_cuts), 0
            for w in self.queue:
                seen += len(w.word)
                hit = bool(cuts) and seen >= cuts[0]
                if hit:
                    cuts.pop(0)
                mask.extend([hit] * len(w.word))
        else:
            for w in self.queue:
                mask.extend([any(w.word.endswith(m) for m in marks)] * len(w.word))
        return mask

    def _reset(self):
        self.seek = 0
        self.time_offset = 0.0
        self.pending: Optional[TimedWord] = None       # AlignmentTmpData.word / .extra_words / .mask / .offset
        self.pending_extra: Optional[List[TimedWord]] = None
        self.pending_mask = None
        self.pending_offset: Optional[float] = None
        self.curr: List[TimedWord] = []
        self.window: List[WordToken] = []
        self.preds: dict = {}

    # ------------------------------------------------------------------------------------------ one window
    def _take_window_words(self) -> Tuple[List[WordToken], List[int], bool]:
        """Pop words for the next window until ``token_step`` tokens (word tokens + one padding token per sentence
        end inside the batch) would be exceeded (:731-750)."""
        m = self.ends_sentence
        starts_after_gap = True
        if m:
            starts_after_gap = True if self.chars_left == len(m) else m[-(self.chars_left + 1)]
        words: List[WordToken] = []
        cuts: List[int] = []
        n_tok = 0
        for i in range(len(self.queue)):
            w = self.queue[0]
            closes = bool(m) and m[-(self.chars_left - len(w.word) + 1)]
            if n_tok + len(cuts) + len(w.tokens) + (1 if closes else 0) > self.token_step and words:
                break
            if closes:
                cuts.append(i + 1)
            self.chars_left -= len(w.word)
            words.append(self.queue.pop(0))
            n_tok += len(w.tokens)
        return words, cuts, starts_after_gap

    def _with_gap_padding(self, words: List[WordToken], cuts: List[int], pad_first: bool) -> List[WordToken]:
        """Insert the ' ...' pseudo-word in front of every sentence of the window (:669-679, 1098-1127)."""
        if not cuts:
            return words
        edges = [0] + cuts
        if edges[-1] < len(words):
            edges.append(len(words))
        groups = [words[a:b] for a, b in zip(edges[:-1], edges[1:])]
        if self.gap_padding is None:
            return [w for g in groups for w in g]
        pad_tokens = self.encode(self.gap_padding)
        pad = WordToken(self.gap_padding, pad_tokens, True)
        n = len(pad_tokens)
        for gi, g in enumerate(groups):
            first, prev_last = g[0].tokens, (groups[gi - 1][-1].tokens if gi else None)
            if (n <= len(first) and pad_tokens == first[:n]) or \
                    (prev_last is not None and n <= len(prev_last) and pad_tokens == prev_last[-n:]) or \
                    (gi == 0 and not pad_first):
                continue
            g.insert(0, pad)
        return [w for g in groups for w in g]

    def _infer(self, audio_segment: torch.Tensor, words: List[WordToken], cuts: List[int], pad_first: bool,
               time_offset: Optional[float] = None) -> List[TimedWord]:
        """Call the inference function for one window (:657-729)."""
        asked = self._with_gap_padding(words, cuts, pad_first)
        got = self.inference_func(audio_segment, asked)
        return self._assemble(asked, got, audio_segment.size(-1), self.time_offset if time_offset is None else time_offset)

    def _assemble(self, asked: List[WordToken], got: List[dict], n_samples: int, time_offset: float) -> List[TimedWord]:
        """Re-assemble the (possibly finer) output of the inference function into the requested words (:680-729);
        padding pseudo-words are dropped, times are clipped to the segment and shifted to absolute."""
        t_max = round(n_samples / self.sample_rate, 4)
        if len(got) < len(asked):
            raise RuntimeError(f"expected output word count to be at least {len(asked)} but got {len(got)}")
        if got[-1]["start"] > t_max:
            warnings.warn(f'word "{got[-1]}" start later than the max timestamp')
        out: List[TimedWord] = []
        k, text, t0, probs = 0, "", -1, []
        for gi, g in enumerate(got):
            text += g["word"]
            if t0 == -1:
                t0 = g["start"]
            if g.get("probability"):
                probs.append(g["probability"])
            want = asked[k]
            if text == want.word:
                if not want.is_padding:
                    a, b = min(t0, t_max), min(g["end"], t_max)
                    out.append(TimedWord(want.word, round3(a + time_offset), round3(b + time_offset), want.tokens,
                                         np.mean(probs).item() if probs else 0.0))
                k, text, t0, probs = k + 1, "", -1, []
            elif len(text) > len(want.word) or gi == len(got) - 1:
                raise RuntimeError(f'expect word "{want.word}" but got "{text}"')
        return out

    # ------------------------------------------------------------------------------- re-alignment bookkeeping
    def _speech_fraction(self, w: TimedWord, mask, offset: float) -> float:
        if mask is None:
            return 1
        a, b = int((w.start - offset) * self.tokens_per_sec), int((w.end - offset) * self.tokens_per_sec)
        return 1 - mask[a:b].float().mean().nan_to_num().item()

    def _keep_new(self, new: TimedWord, new_mask, new_off: float, old: TimedWord, old_mask, old_off: float) -> bool:
        """:803-816 -- True when the re-timed word (`new`, from the current window) should replace the earlier timing
        (`old`): it is not much less confident and lies at least as much in speech, or it is simply more confident."""
        s_new = round(self._speech_fraction(new, new_mask, new_off), 1)
        s_old = round(self._speech_fraction(old, old_mask, old_off), 1)
        return ((old.probability ** 0.75 - new.probability ** 0.75) < 0.35 and s_new >= s_old) or \
            new.probability >= old.probability

    def _rejoin(self, target: TimedWord, pieces: List[TimedWord], second: Optional[TimedWord] = None):
        """:752-788 -- the earlier window may have timed `target` as several pieces (it was cut at a different word
        boundary): glue pieces until they spell `target`; returns (glued word or None, remaining pieces)."""
        head = pieces[0]
        assert target.word.startswith(head.word)
        if target.word != head.word:
            if len(pieces) < 2:
                return None, []
            probs = [head.probability]
            if head.word.strip() in self.all_punctuations:
                head.start, head.end = pieces[1].start, pieces[1].end
            for _ in range(len(pieces) - 1):
                nxt = pieces.pop(1)
                joined = head.word + nxt.word
                assert target.word.startswith(joined)
                head.word = joined
                head.tokens += nxt.tokens
                probs.append(nxt.probability)
                if nxt.word.strip() not in self.all_punctuations:
                    head.end = nxt.end
                if target.word == head.word:
                    break
            if target.word != head.word:
                return None, []
            head.probability = np.mean(probs).item()
        elif second:
            if len(pieces) == 1:
                return head, []
            nxt, rest = self._rejoin(second, pieces[1:])
            if nxt is not None:
                rest = [nxt] + rest
            return head, rest
        return head, pieces[1:]

    def _adopt_pending(self):
        """:818-823 -- the pending word (and the timings that followed it) overwrite the head of the current window."""
        if self.pending is None:
            return
        keep = [self.pending] + self.pending_extra[:len(self.curr) - 1]
        self.curr[:len(keep)] = keep
        self.pending = None

    def _requeue(self, index: Optional[int]):
        """:825-871 -- decide between old and new timings of the overlap word(s), then push the untrusted tail of the
        window (from `index`; everything when None) back to the front of the queue.  The last trusted word becomes
        the new pending word: it goes back on the queue too and opens the next window."""
        if index is not None and self.curr and self.pending is not None:
            self.pending, self.pending_extra = self._rejoin(
                self.curr[0], [self.pending] + self.pending_extra, self.curr[1] if len(self.curr) > 1 else None)
            if self.pending:
                kept: List[TimedWord] = []
                if self._keep_new(self.curr[0], self.preds["mask"], self.time_offset,
                                  self.pending, self.pending_mask, self.pending_offset):
                    self.pending = self.curr[0]
                else:
                    for cw, ow in zip(self.curr[1:], self.pending_extra):
                        assert cw.word.startswith(ow.word)
                        if self._keep_new(cw, self.preds["mask"], self.time_offset, ow, self.pending_mask,
                                          self.pending_offset) or cw.word != ow.word or cw.end < ow.end:
                            break
                        kept.append(ow)
                self.pending_extra = kept
        if index is None:
            self.chars_left += sum(len(w.word) for w in self.window)
            self.queue = self.window + self.queue
            self.curr = []
            self.pending = None
        elif index != len(self.window):
            self.chars_left += sum(len(w.word) for w in self.window[index:])
            self.queue = self.window[index:] + self.queue
            self.curr, tail = self.curr[:index], self.curr[index:]
            if self.curr:
                self._adopt_pending()
                self.chars_left += sum(len(w.word) for w in self.window[index - 1:index])
                self.queue = self.window[index - 1:index] + self.queue
                self.pending = self.curr.pop(-1)
                self.pending_extra = tail
                self.pending_mask = self.preds["mask"]
                self.pending_offset = self.time_offset
        else:
            self._adopt_pending()

    def _settle_window(self, segment_samples: int) -> float:
        """:937-1006 -- which words of this window are trusted, and where does the next window start?
        Words with zero duration at the tail are untrusted; a last word touching the end of the window is re-done
        with more context; implausibly long words (vs ``word_dur_factor`` x the window's median duration and
        ``max_word_dur``) either move the next start backwards in front of them or cut the trusted run short."""
        dur = np.array([w.end - w.start for w in self.curr]).round(3)
        good = dur > 0
        idx = np.flatnonzero(good)
        if not len(idx):
            self.seek += segment_samples
            last_ts = round(self.seek / self.sample_rate, 2)
            self._requeue(None)
            return last_ts
        redo = idx[-1] + 1
        if self.queue and len(idx) > 1 and \
                self.curr[idx[-1]].end >= np.floor(self.time_offset + segment_samples / self.sample_rate):
            good[idx[-1]] = False
            idx = idx[:-1]
            redo = idx[-1] + 1
        med = np.median(dur[:redo])
        new_start = None
        cap_all = None
        if not self.fast_mode:
            cap_here = round(med * self.word_dur_factor, 3) if self.word_dur_factor else None
            if self.max_word_dur:
                cap_here = min(cap_here, self.max_word_dur) if cap_here else self.max_word_dur
                cap_all = self.max_word_dur
            else:
                cap_all = cap_here or None
            if cap_all and med > cap_all:
                med = cap_all
            if cap_here and dur[idx[0]] > cap_all:
                first = self.curr[idx[0]]
                new_start = round(max(first.end - (med * idx[0] + cap_here), first.start), 3)
                if new_start <= self.time_offset:
                    new_start = None
        if new_start is None:
            if cap_all:
                lo = idx[0] + 1
                too_long = np.flatnonzero(dur[lo:redo] > cap_all) + lo
                if len(too_long):
                    redo = too_long[0]
            last_ts = self.curr[redo - 1].end
            self._requeue(int(redo))
        else:
            last_ts = new_start
            self._requeue(None)
        self.seek = round(last_ts * self.sample_rate)
        return last_ts

    def _skip_nonspeech(self, segment: torch.Tensor) -> Optional[torch.Tensor]:
        """:873-935 -- jump over a detected non-speech stretch of at least ``nonspeech_skip`` seconds at the start of
        the window, and stop the (re-loaded) window where the next such stretch begins."""
        if self.nonspeech_skip is None:
            return segment
        t = self.preds["timings"]
        if t is None or len(t[0]) == 0:
            return segment
        n = segment.size(-1)
        hi, lo = self.time_offset + self.min_word_dur, self.time_offset - self.min_word_dur
        if t[0][0] < hi and t[1][0] > lo + n / self.sample_rate:
            self.seek += n                               # the whole window lies inside the first non-speech section
            return None
        long_enough = (t[1] - t[0]) >= self.nonspeech_skip
        if not long_enough.any():
            return segment
        starts = t[0, long_enough]
        if hi < starts[0]:
            return segment                               # speech comes first
        ends = t[1, long_enough]
        total = self.audio.get_total_samples()
        self.seek = round(ends[0] * self.sample_rate)
        if self.seek + self.min_word_dur * self.sample_rate > total:
            self.seek = total
            return None
        self.time_offset = self.seek / self.sample_rate
        segment = self.audio.next_chunk(self.seek, self.n_samples)
        if segment is None:
            return None
        self.preds = self.detector.predict(segment, offset=self.time_offset)
        if len(starts) > 1:
            segment = segment[:round((starts[1] - ends[0]) * self.sample_rate)]
        return segment

    # ------------------------------------------------------------------------------------------------- driver
    def align(self, audio: torch.Tensor, text) -> Optional[WhisperResult]:
        self._reset()
        self._load_text(text)
        self.audio = _MemoryAudio(audio, self.sample_rate)
        # suppress_silence=False keeps only the exact-zero-sample mask (stabilization/__init__.py:84-87, 271-286)
        self.detector = NonSpeechPredictor(q_levels=self.q_levels, k_size=self.k_size, min_word_dur=self.min_word_dur,
                                           min_silence_dur=self.min_silence_dur, get_mask=True, pad_mask=False,
                                           loudness=self.suppress_silence)
        done: List[TimedWord] = []
        last_ts = 0.0
        while self.queue:
            self.time_offset = self.seek / self.sample_rate
            segment = self.audio.next_chunk(self.seek, self.n_samples)
            if segment is None:
                break
            self.preds = self.detector.predict(segment, offset=self.time_offset)
            segment = self._skip_nonspeech(segment)
            if segment is None:
                continue
            self.curr = self._infer(segment, *self._take_window_words())
            self.window = [WordToken(w.word, w.tokens) for w in self.curr]
            last_ts = self._settle_window(segment.shape[-1])
            if self.progress_callback is not None:
                self.progress_callback(min(round(last_ts, 2), self.audio.get_duration(2)), self.audio.get_duration(2))
            done.extend(self.curr)
            if self.failure_threshold is not None:
                self.failure_count += sum(1 for w in self.curr if w.end - w.start == 0)
                if self.failure_count > self.max_fail:
                    break
        if self.pending is not None:
            done.append(self.pending)
        if not done:
            warnings.warn("Failed to align text.", stacklevel=2)
        if self.failure_count > self.max_fail:
            warnings.warn(f"Alignment aborted. Failed word percentage exceeded {self.failure_threshold * 100}% at "
                          f"{self.seek / self.sample_rate:.3f}s.", stacklevel=2)
        elif self.queue:
            warnings.warn(f"Failed to align the last {len(self.queue)}/{self.total_words} words after "
                          f"{(done[-1].end if done else 0):.3f}s.", stacklevel=2)
        if self.queue and not self.remove_instant_words:
            eof = self.audio.get_duration(3)
            done.extend(TimedWord(w.word, eof, eof, w.tokens, 0.0) for w in self.queue)
        if not done:
            return None
        words = [dict(word=w.word, start=w.start, end=w.end, tokens=w.tokens, probability=w.probability) for w in done]
        if len(self._char_cuts):
            lens = np.cumsum([len(w.word) for w in done])
            stops = [int(np.flatnonzero(lens >= c)[0]) + 1 for c in self._char_cuts]
            result = WhisperResult([words[a:b] for a, b in zip([0] + stops[:-1], stops) if a != b])
        else:
            result = WhisperResult([words])
        timings = self.detector.timings()
        if self.suppress_silence and timings is not None:
            result.suppress_silence(*timings, min_word_dur=self.min_word_dur, word_level=self.suppress_word_ts,
                                    nonspeech_error=self.nonspeech_error, use_word_position=self.use_word_position)
            result.update_nonspeech_sections(*timings)
            result.set_current_as_orig()
        if not self.original_split:
            result.regroup(self.regroup)
        n_bad = sum(1 for s in result.segments if s.end - s.start <= 0)
        if n_bad:
            warnings.warn(f"{n_bad}/{len(result.segments)} segments failed to align.", stacklevel=2)
        return result

    # ------------------------------------------------------------------------------------------- align_words
    def align_words(self, audio: torch.Tensor, result, normalize_text: bool = True, inplace: bool = True,
                    batch_inference: Optional[Callable] = None, batch_size: int = 1) -> WhisperResult:
        """:396-474 -- every segment keeps its start/end and only its words are (re-)timed inside that span; no gap
        padding, no fallback, segments are independent.  Because they are independent, ``batch_inference(list of
        segments, list of word lists)`` may time ``batch_size`` of them per device pass; the results are the same."""
        import copy
        self._reset()
        seg_tokens = None
        if isinstance(result, WhisperResult):
            if not inplace:
                result = copy.deepcopy(result)
        else:
            if result and not result[0].get("text") and result[0].get("tokens"):
                seg_tokens = [list(s["tokens"]) for s in result]
                result = [dict(s, text=self.decode(s["tokens"])) for s in result]
            result = WhisperResult(result)

        def norm(text: str) -> str:
            if not normalize_text:
                return text
            text = re.sub(r"\s", " ", text)
            return text if text.startswith(" ") else " " + text

        if seg_tokens is None:
            seg_tokens = [self.encode(norm(s.text)) for s in result]
        too_long = [i for i, t in enumerate(seg_tokens) if len(t) > self.token_step]
        if too_long:
            raise RuntimeError(f"found segments at following indices exceeding max length for model: {too_long}")
        self.audio = _MemoryAudio(audio, self.sample_rate)
        self.detector = NonSpeechPredictor(q_levels=self.q_levels, k_size=self.k_size, min_word_dur=self.min_word_dur,
                                           min_silence_dur=self.min_silence_dur, get_mask=True, pad_mask=False,
                                           loudness=self.suppress_silence)
        jobs = []
        for seg, toks in zip(result.segments, seg_tokens):
            if seg.duration == 0:
                continue
            chunk = self.audio.next_chunk(round(seg.start * self.sample_rate), round(seg.duration * self.sample_rate))
            if chunk is None:
                break
            self.detector.predict(chunk, offset=seg.start)
            words = tokens_to_word_tokens(toks, self.decode, self.split_words_by_space)
            jobs.append((seg, chunk, words))
        step = max(int(batch_size), 1) if batch_inference is not None else 1
        for k in range(0, len(jobs), step):
            part = jobs[k:k + step]
            if batch_inference is not None:
                outs = batch_inference([c for _, c, _ in part], [w for _, _, w in part])
            else:
                outs = [self.inference_func(c, w) for _, c, w in part]
            for (seg, chunk, words), got in zip(part, outs):
                timed = self._assemble(words, got, chunk.size(-1), seg.start)
                seg.words = [WordTiming(w.word, w.start, w.end, w.probability, w.tokens) for w in timed]
        result.reassign_ids()
        timings = self.detector.timings()
        if self.suppress_silence and timings is not None:
            result.suppress_silence(*timings, min_word_dur=self.min_word_dur, word_level=self.suppress_word_ts,
                                    nonspeech_error=self.nonspeech_error, use_word_position=self.use_word_position)
            result.update_nonspeech_sections(*timings)
            result.set_current_as_orig()
        result.regroup(self.regroup)
        return result
