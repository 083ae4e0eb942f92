"""Result containers returned by transcribe()/align(): WordTiming, Segment, WhisperResult.

Mirrors the reference's result model where consumers touch it (stable_whisper/result.py: WordTiming :74-257, Segment
:277-925, WhisperResult :928-3102): the same field names, the same ``to_dict()`` JSON schema (:192-204, :596-626,
:1398-1406), millisecond rounding of every stored timestamp (:38-41), lock flags, ordering checks, and the regrouping
methods (``split_by_*``, ``merge_by_*``, ``clamp_max``, ``lock``, ``pad``, ``regroup`` and its string DSL).  The
regrouping algorithms themselves live in :mod:`stable_ts_amd.regroup` as functions over the segment list; the methods
here are the reference-named entry points that record ``regroup_history``.

Pure host-side list surgery after the GPU hot path (SURVEY.md 8f "next-1"); there is nothing here for the GPU to do.
"""
import json
import warnings
from typing import Iterator, List, Optional, Union

from ._num import round3


def _ms(ts):
    """result.py:38-41: timestamps are stored rounded to the millisecond (0 / None pass through)."""
    return round3(ts) if ts else ts


def _blend(a, b):
    """result.py:21-31 for scalars: merged-segment statistics are the mean of the pair, None is contagious."""
    if a is None:
        return None
    return None if b is None else (a + b) / 2


def format_timestamp(seconds: float, always_include_hours: bool = False, decimal_marker: str = ".") -> str:
    """utils.py:47-65: ``[HH:]MM:SS.mmm``"""
    assert seconds >= 0, "non-negative timestamp expected"
    ms = round(seconds * 1000.0)
    hours, ms = divmod(ms, 3_600_000)
    minutes, ms = divmod(ms, 60_000)
    secs, ms = divmod(ms, 1_000)
    head = f"{hours:02d}:" if always_include_hours or hours > 0 else ""
    return f"{head}{minutes:02d}:{secs:02d}{decimal_marker}{ms:03d}"


def group_by_lock(words: List["WordTiming"], only_text: bool = False, include_single: bool = False) -> list:
    """result.py:260-274: runs of words that lock flags tie together (a boundary is tied when the left word is
    right-locked or the right word is left-locked)."""
    runs: List[list] = []
    for w in words:
        if runs and (runs[-1][-1].right_locked or w.left_locked):
            runs[-1].append(w)
        else:
            runs.append([w])
    if not include_single:
        runs = [r for r in runs if len(r) > 1]
    return [[w.word for w in r] for r in runs] if only_text else runs


def _deprecated(what: str, instead: str):
    warnings.warn(f"``{what}`` is deprecated and will be removed in future versions. {instead}", stacklevel=3)


class WordTiming:
    __slots__ = ("word", "_start", "_end", "probability", "tokens", "left_locked", "right_locked", "id", "segment",
                 "round_ts")

    def __init__(self, word: str, start: float, end: float, probability: Optional[float] = None,
                 tokens: Optional[List[int]] = None, left_locked: bool = False, right_locked: bool = False,
                 segment_id: Optional[int] = None, id: Optional[int] = None, segment: Optional["Segment"] = None,
                 round_ts: bool = True, **_):
        self.round_ts = round_ts
        self.word = word
        self._start = self.round(start)
        self._end = self.round(end)
        self.probability = probability
        self.tokens = tokens
        self.left_locked = bool(left_locked)
        self.right_locked = bool(right_locked)
        self.id = id
        self.segment = segment

    # -- timestamps (rounded on every store, result.py:160-172)
    @property
    def start(self):
        return self._start

    @start.setter
    def start(self, v):
        self._start = self.round(v)

    @property
    def end(self):
        return self._end

    @end.setter
    def end(self, v):
        self._end = self.round(v)

    @property
    def duration(self):
        return self.round(self._end - self._start)

    def round(self, timestamp):
        return _ms(timestamp) if self.round_ts else timestamp

    def round_all_timestamps(self):
        _deprecated(".round_all_timestamps()", "Use ``.round_ts=True`` to round timestamps by default instead.")
        self.round_ts = True

    def to_display_str(self) -> str:
        return f'[{format_timestamp(self.start)}] -> [{format_timestamp(self.end)}] "{self.word}"'

    def set_segment(self, segment: "Segment"):
        _deprecated(".set_segment(current_segment_instance)", "Use ``.segment = current_segment`` instead.")
        self.segment = segment

    def get_segment(self) -> Optional["Segment"]:
        warnings.warn("``.get_segment()`` will be removed in future versions. Use ``.segment`` instead.", stacklevel=2)
        return self.segment

    @property
    def segment_id(self):
        return None if self.segment is None else self.segment.id

    def __len__(self):
        return len(self.word)

    def __repr__(self):
        return f'WordTiming(start={self.start}, end={self.end}, word="{self.word}")'

    def copy(self, keep_segment: bool = False, copy_tokens: bool = False) -> "WordTiming":
        toks = self.tokens
        if toks is not None and copy_tokens:
            toks = list(toks)
        return WordTiming(self.word, self._start, self._end, self.probability, toks, self.left_locked,
                          self.right_locked, id=self.id, segment=self.segment if keep_segment else None,
                          round_ts=self.round_ts)

    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo=None):
        return self.copy(copy_tokens=True)

    def joined(self, other: "WordTiming") -> "WordTiming":
        """result.py:111-127 (``a + b``): text concatenated, span = union, probability averaged, tokens chained."""
        w = WordTiming(self.word + other.word, min(self.start, other.start), max(self.end, other.end),
                       _blend(self.probability, other.probability),
                       None if self.tokens is None or other.tokens is None else list(self.tokens) + list(other.tokens),
                       self.left_locked or other.left_locked, self.right_locked or other.right_locked,
                       id=self.id, segment=self.segment)
        return w

    __add__ = joined

    def offset_time(self, offset: float):
        self.start = self._start + offset
        self.end = self._end + offset

    def rescale_time(self, factor: float):
        self.start = self._start * factor
        self.end = self._end * factor

    def lock_left(self):
        self.left_locked = True

    def lock_right(self):
        self.right_locked = True

    def lock_both(self):
        self.left_locked = self.right_locked = True

    def unlock_both(self):
        self.left_locked = self.right_locked = False

    def suppress_silence(self, silent_starts, silent_ends, min_word_dur: Optional[float] = None,
                         nonspeech_error: float = 0.3, keep_end: Optional[bool] = True):
        from .stabilization import snap_to_speech
        snap_to_speech(self, silent_starts, silent_ends, 0.1 if min_word_dur is None else min_word_dur, nonspeech_error,
                       keep_end)
        return self

    def clamp_max(self, max_dur: float, clip_start: bool = False):
        """result.py:231-244: shorten an over-long word from one side."""
        if self.duration > max_dur:
            if clip_start:
                self.start = round(self._end - max_dur, 3)
            else:
                self.end = round(self._start + max_dur, 3)

    def to_dict(self) -> dict:
        return dict(word=self.word, start=self.start, end=self.end, probability=self.probability,
                    tokens=None if self.tokens is None else list(self.tokens))


class Segment:
    def __init__(self, start: Optional[float] = None, end: Optional[float] = None, text: Optional[str] = None,
                 seek: Optional[float] = None, tokens: Optional[List[int]] = None, temperature: Optional[float] = None,
                 avg_logprob: Optional[float] = None, compression_ratio: Optional[float] = None,
                 no_speech_prob: Optional[float] = None, words: Optional[List[Union[WordTiming, dict]]] = None,
                 id: Optional[int] = None, result: Optional["WhisperResult"] = None, round_ts: bool = True, **_):
        if words:                                   # with words, start/end/text/tokens are views of the words
            start = end = text = tokens = None
        self.round_ts = round_ts
        self._default_start = self.round(start) if start else 0.0
        self._default_end = self.round(end) if end else 0.0
        self._default_text = text or ""
        self._default_tokens = tokens or []
        self.seek = seek
        self.temperature = temperature
        self.avg_logprob = avg_logprob
        self.compression_ratio = compression_ratio
        self.no_speech_prob = no_speech_prob
        self.id = id
        self.result = result
        self.words: Optional[List[WordTiming]] = None
        if words is not None:
            self.words = [w if isinstance(w, WordTiming) else WordTiming(**w, round_ts=round_ts) for w in words]
            self.reassign_ids()

    # -- views
    @property
    def has_words(self) -> bool:
        return bool(self.words)

    @property
    def ori_has_words(self) -> bool:
        return self.words is not None

    @property
    def start(self):
        return self.words[0].start if self.words else self._default_start

    @start.setter
    def start(self, v):
        if self.words:
            self.words[0].start = v
        else:
            self._default_start = self.round(v)

    @property
    def end(self):
        return self.words[-1].end if self.words else self._default_end

    @end.setter
    def end(self, v):
        if self.words:
            self.words[-1].end = v
        else:
            self._default_end = self.round(v)

    @property
    def text(self) -> str:
        return "".join(w.word for w in self.words) if self.words else self._default_text

    @property
    def tokens(self) -> List[int]:
        if self.words and self.words[0].tokens:
            return [t for w in self.words for t in w.tokens]
        return self._default_tokens

    @property
    def duration(self):
        return self.end - self.start

    @property
    def left_locked(self) -> bool:
        return bool(self.words) and self.words[0].left_locked

    @property
    def right_locked(self) -> bool:
        return bool(self.words) and self.words[-1].right_locked

    def word_count(self) -> int:
        return len(self.words) if self.words else -1

    def char_count(self) -> int:
        return sum(len(w.word) for w in self.words) if self.words else len(self.text)

    def __getitem__(self, i) -> WordTiming:
        if self.words is None:
            raise ValueError("segment contains no words")
        return self.words[i]

    def __delitem__(self, i):
        if self.words is None:
            raise ValueError("segment contains no words")
        del self.words[i]
        self.reassign_ids(i)

    def __repr__(self):
        return f'Segment(start={self.start}, end={self.end}, text="{self.text}")'

    def round(self, timestamp):
        return _ms(timestamp) if self.round_ts else timestamp

    def round_all_timestamps(self):
        _deprecated(".round_all_timestamps()", "Use ``.round_ts=True`` to round timestamps by default instead.")
        self.round_ts = True

    def to_display_str(self, only_segment: bool = False) -> str:
        line = f'[{format_timestamp(self.start)} --> {format_timestamp(self.end)}] "{self.text}"'
        if self.words and not only_segment:
            line += "\n" + "\n".join(f"-{w.to_display_str()}" for w in self.words) + "\n"
        return line

    def set_result(self, result: "WhisperResult"):
        _deprecated(".set_result(current_result_instance)", "Use ``.result = current_result_instance`` instead.")
        self.result = result

    def get_result(self) -> Optional["WhisperResult"]:
        warnings.warn("``.get_result()`` will be removed in future versions. Use ``.result`` instead.", stacklevel=2)
        return self.result

    def update_seg_with_words(self):
        _deprecated("update_seg_with_words()", "Use ``.reassign_ids()`` to manually update ids")
        self.reassign_ids()

    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo=None):
        return self.copy(copy_words=True, copy_tokens=True)

    # -- word-level surgery and the index pickers of the regrouping methods (result.py:464-560, 707-902)
    def add(self, other: "Segment", copy_words: bool = False, newline: bool = False, reassign_ids: bool = True) -> "Segment":
        """the two segments fused into a new one: words concatenated, decode statistics averaged (:464-493)"""
        from .regroup import _fuse
        if self.ori_has_words != other.ori_has_words:
            a, b = ("with" if s.ori_has_words else "without" for s in (self, other))
            raise ValueError(f"Can't merge segment {a} words and a segment {b} words.")
        left, right = (self.copy(copy_words=True), other.copy(copy_words=True)) if copy_words else (self, other)
        fused = _fuse(left, right, newline)
        if reassign_ids:
            fused.reassign_ids()
        return fused

    def __add__(self, other: "Segment") -> "Segment":
        return self.add(other, copy_words=True)

    def add_words(self, index0: int, index1: int, inplace: bool = False) -> Optional[WordTiming]:
        if not self.words:
            return None
        joined = self.words[index0] + self.words[index1]
        if inplace:
            i0, i1 = sorted([index0, index1])
            self.words[i0] = joined
            del self.words[i1]
        return joined

    def apply_min_dur(self, min_dur: float, inplace: bool = False) -> "Segment":
        """Words shorter than ``min_dur`` are joined with a neighbour -- the shorter one when both exist (:536-560)."""
        seg = self if inplace else self.copy(copy_words=True, copy_tokens=True)
        if not self.words or len(seg.words) < 2:
            return seg
        last = len(seg.words) - 1
        for i in reversed(range(len(seg.words))):
            if last == 0:
                break
            if seg.words[i].duration < min_dur:
                if i == last:
                    seg.add_words(i - 1, i, inplace=True)
                elif i == 0:
                    seg.add_words(i, i + 1, inplace=True)
                elif seg.words[i + 1].duration < seg.words[i - 1].duration:
                    seg.add_words(i - 1, i, inplace=True)
                else:
                    seg.add_words(i, i + 1, inplace=True)
                last -= 1
        return seg

    def words_by_lock(self, only_text: bool = True, include_single: bool = False) -> list:
        return group_by_lock(self.words, only_text=only_text, include_single=include_single)

    def get_locked_indices(self) -> List[int]:
        from .regroup import _locked_cuts
        return _locked_cuts(self.words)

    def get_gaps(self, as_ndarray: bool = False):
        import numpy as np
        if not self.words:
            return []
        gaps = np.array([w.start for w in self.words])[1:] - np.array([w.end for w in self.words])[:-1]
        return gaps if as_ndarray else gaps.tolist()

    def get_gap_indices(self, max_gap: float = 0.1) -> List[int]:
        from .regroup import gap_cuts
        return gap_cuts(self, max_gap)

    def get_punctuation_indices(self, punctuation) -> List[int]:
        from .regroup import punctuation_cuts
        return punctuation_cuts(self, punctuation)

    def get_length_indices(self, max_chars: int = None, max_words: int = None, even_split: bool = True,
                           include_lock: bool = False, ignore_special_periods: bool = False) -> List[int]:
        from .regroup import length_cuts
        return [int(i) for i in length_cuts(self, max_chars, max_words, even_split, include_lock, ignore_special_periods)]

    def get_duration_indices(self, max_dur: float, even_split: bool = True, include_lock: bool = False,
                             ignore_special_periods: bool = False) -> List[int]:
        from .regroup import duration_cuts
        return [int(i) for i in duration_cuts(self, max_dur, even_split, include_lock, ignore_special_periods)]

    def split(self, indices: List[int], reassign_ids: bool = True) -> List["Segment"]:
        """pieces that end after the words ``indices`` (:886-902); the words are shared, not copied"""
        if len(indices) == 0:
            return []
        from .regroup import _pieces
        parts = _pieces(self, [i for i in indices if i != len(self.words) - 1])
        if reassign_ids:
            for p in parts:
                p.reassign_ids()
        return parts

    # -- construction helpers used by the regrouping functions
    def spawn(self, words: Optional[List[WordTiming]]) -> "Segment":
        """A segment that inherits this one's decode statistics and owns ``words`` (result.py:356-389 with new_words)."""
        s = Segment(seek=self.seek, temperature=self.temperature, avg_logprob=self.avg_logprob,
                    compression_ratio=self.compression_ratio, no_speech_prob=self.no_speech_prob, id=self.id)
        s.words = words
        return s

    def copy(self, new_words: Optional[List[WordTiming]] = None, keep_result: bool = False, copy_words: bool = False,
             copy_tokens: bool = False) -> "Segment":
        """result.py:356-396: same decode statistics; the words are shared unless ``copy_words``; with ``new_words`` the
        copy owns those words instead and drops the segment-level defaults."""
        src = self.words if new_words is None else new_words
        words = None if src is None or (new_words is None and not self.words) else (
            [w.copy(copy_tokens=copy_tokens) for w in src] if copy_words else src)
        s = Segment(seek=self.seek, temperature=self.temperature, avg_logprob=self.avg_logprob,
                    compression_ratio=self.compression_ratio, no_speech_prob=self.no_speech_prob, id=self.id,
                    result=self.result if keep_result else None, round_ts=self.round_ts)
        s.words = words
        if new_words is None:
            s._default_start, s._default_end = self._default_start, self._default_end
            s._default_text, s._default_tokens = self._default_text, self._default_tokens
        return s

    def reassign_ids(self, start: Optional[int] = None):
        if self.words:
            for i, w in enumerate(self.words[start:], start or 0):
                w.segment, w.id = self, i

    def lock_left(self):
        if self.words:
            self.words[0].lock_left()

    def lock_right(self):
        if self.words:
            self.words[-1].lock_right()

    def lock_both(self):
        self.lock_left()
        self.lock_right()

    def unlock_all_words(self):
        for w in self.words or ():
            w.unlock_both()

    def offset_time(self, offset: float):
        if self.seek is not None:
            self.seek += offset
        if self.words:
            for w in self.words:
                w.offset_time(offset)
        else:
            self.start = self.start + offset
            self.end = self.end + offset

    def rescale_time(self, factor: float):
        if self.seek is not None:
            self.seek *= factor
        if self.words:
            for w in self.words:
                w.rescale_time(factor)
        else:
            self.start = self.start * factor
            self.end = self.end * factor

    def suppress_silence(self, silent_starts, silent_ends, min_word_dur: Optional[float] = None, word_level: bool = True,
                         nonspeech_error: float = 0.3, use_word_position: bool = True):
        """result.py:681-705: snap the words (or only the outer two) out of the non-speech sections; a word keeps its
        END fixed unless it closes a clause (ends with closing punctuation) or the segment."""
        from .stabilization import snap_to_speech
        from .timing import APPEND_PUNCTUATIONS
        mwd = 0.1 if min_word_dur is None else min_word_dur
        if self.words:
            sel = self.words if word_level or len(self.words) == 1 else [self.words[0], self.words[-1]]
            for i, w in enumerate(sel, 1):
                keep_end = (not (w.word[-1] in APPEND_PUNCTUATIONS or i == len(sel))) if use_word_position else None
                snap_to_speech(w, silent_starts, silent_ends, mwd, nonspeech_error, keep_end)
        else:
            snap_to_speech(self, silent_starts, silent_ends, mwd, nonspeech_error, True)
        return self

    def convert_to_segment_level(self):
        """result.py:918-925: freeze the word-derived fields and drop the words."""
        if self.words:
            self._default_text, self._default_tokens = self.text, self.tokens
            self._default_start, self._default_end = self.start, self.end
            self.words = None

    def to_dict(self) -> dict:
        toks = self.tokens
        d = dict(start=self.start, end=self.end, text=self.text, seek=self.seek,
                 tokens=None if toks is None else list(toks), temperature=self.temperature,
                 avg_logprob=self.avg_logprob, compression_ratio=self.compression_ratio,
                 no_speech_prob=self.no_speech_prob)
        if self.words:
            d["words"] = [w.to_dict() for w in self.words]
        elif self.words is not None:
            d["words"] = []
        return d


class UnsortedException(Exception):
    pass


class WhisperResult:
    def __init__(self, result: Union[dict, list, str], force_order: bool = False, check_sorted: bool = True):
        self.path = None
        if isinstance(result, str):
            self.path = result
            with open(result, "r", encoding="utf-8") as f:
                result = json.load(f)
        result = self._as_dict(result)
        self.ori_dict = result.get("ori_dict") or result
        self.language = self.ori_dict.get("language")
        self._regroup_history = result.get("regroup_history", "")
        self._nonspeech_sections = result.get("nonspeech_sections") or []
        segs = result.get("segments", self.ori_dict.get("segments")) or []
        self.segments: List[Segment] = [s if isinstance(s, Segment) else Segment(**s) for s in segs]
        self._forced_order = force_order
        self._ignore_special_periods = False
        self.unfinished_start = result.get("unfinished", result.get("unfinished_start", -1.0))
        if force_order:
            self.force_order()
        if check_sorted:
            self.raise_for_unsorted()
        self.remove_no_word_segments(any(s.has_words for s in self.segments))

    @staticmethod
    def _as_dict(result) -> dict:
        """result.py:966-996: accept a dict, a list of segment dicts, or a list of word-dict lists."""
        if isinstance(result, dict):
            return result
        if not isinstance(result, list):
            raise TypeError(f"Expect result to be list but got {type(result)}")
        if not result or not result[0]:
            return {}
        if isinstance(result[0], list):
            return dict(segments=[dict(start=ws[0]["start"], end=ws[-1]["end"], text="".join(w["word"] for w in ws),
                                       words=ws) for ws in result if ws])
        if isinstance(result[0], (dict, Segment)):
            return dict(segments=result)
        raise NotImplementedError(f"Got list of {type(result[0])} but expects list of list/dict")

    # -- container protocol
    def __len__(self):
        return len(self.segments)

    def __getitem__(self, i) -> Segment:
        return self.segments[i]

    def __delitem__(self, i):
        del self.segments[i]
        self.reassign_ids(True, start=i)

    def __iter__(self) -> Iterator[Segment]:
        return iter(self.segments)

    def __bool__(self):
        return len(self.segments) != 0

    @property
    def has_words(self) -> bool:
        return bool(self.segments) and all(s.has_words for s in self.segments)

    @property
    def text(self) -> str:
        return "".join(s.text for s in self.segments)

    @property
    def duration(self):
        return _ms(self.segments[-1].end - self.segments[0].start) if self.segments else 0.0

    @property
    def regroup_history(self) -> str:
        return self._regroup_history

    @property
    def nonspeech_sections(self) -> list:
        return self._nonspeech_sections

    @nonspeech_sections.setter
    def nonspeech_sections(self, v):
        self._nonspeech_sections = list(v or [])

    def update_nonspeech_sections(self, silent_starts, silent_ends, overwrite: bool = True):
        secs = [dict(start=round3(s), end=round3(e)) for s, e in zip(silent_starts, silent_ends)]     # (= round(x, 3), _num.py)
        if overwrite:
            self._nonspeech_sections = secs
        else:
            self._nonspeech_sections.extend(secs)

    def all_words(self) -> List[WordTiming]:
        return [w for s in self.segments for w in (s.words or [])]

    def all_words_or_segments(self):
        return self.all_words() if self.has_words else self.segments

    def all_tokens(self) -> List[int]:
        if self.has_words:
            return [t for w in self.all_words() for t in (w.tokens or [])]
        return [t for s in self.segments for t in (s.tokens or [])]

    def reassign_ids(self, only_segments: bool = False, start: Optional[int] = None):
        for i, s in enumerate(self.segments[start:], start or 0):
            s.id, s.result = i, self
            if not only_segments:
                s.reassign_ids()

    def remove_no_word_segments(self, ignore_ori: bool = False, reassign_ids: bool = True):
        """result.py:1334-1339: segments that were given a (now empty) word list are dropped."""
        self.segments = [s for s in self.segments if not ((ignore_ori or s.ori_has_words) and not s.has_words)]
        if reassign_ids:
            self.reassign_ids()

    def force_order(self):
        """result.py:998-1018: clip starts to the previous end so the sequence is non-decreasing."""
        parts = self.all_words_or_segments()
        prev_end = 0
        for i, p in enumerate(parts, 1):
            if p.start < prev_end:
                p.start = prev_end
            if p.start > p.end:
                if prev_end > p.end:
                    warnings.warn("Multiple consecutive timestamps are out of order. Some parts will have no duration.")
                    p.start = p.end
                    for q in reversed(parts[:i - 1]):
                        if q.end > p.end:
                            q.end = p.end
                        if q.start > p.end:
                            q.start = p.end
                elif p.start != prev_end:
                    p.start = prev_end
                else:
                    p.end = p.start if i == len(parts) else parts[i].start
            prev_end = p.end

    def raise_for_unsorted(self):
        """result.py:1020-1056: every timestamp must be non-decreasing."""
        stamps = []
        for p in self.all_words_or_segments():
            stamps.extend((p.start, p.end))
        for a, b in zip(stamps[:-1], stamps[1:]):
            if b < a:
                raise UnsortedException(f"timestamps are not in ascending order: {a} -> {b}")

    def offset_time(self, offset: float):
        for s in self.segments:
            s.offset_time(offset)

    def rescale_time(self, factor: float):
        for s in self.segments:
            s.rescale_time(factor)

    def update_all_segs_with_words(self):
        _deprecated("update_all_segs_with_words()", "Use ``.reassign_ids()`` to manually update ids")
        self.reassign_ids()

    def __copy__(self):
        return self.__deepcopy__()

    def __deepcopy__(self, memo=None):
        import copy
        other = WhisperResult.__new__(WhisperResult)
        for k, v in self.__dict__.items():
            if k != "segments":
                setattr(other, k, copy.deepcopy(v, memo))
        other.segments = [sg.copy(copy_words=True, copy_tokens=True) for sg in self.segments]
        other.reassign_ids()
        return other

    def apply_min_dur(self, min_dur: float, inplace: bool = False) -> "WhisperResult":
        """Segments, then words, shorter than ``min_dur`` are merged into a neighbour -- the shorter one when both exist
        (result.py:1106-1131)."""
        res = self if inplace else self.__deepcopy__()
        last = len(res.segments) - 1
        if last == 0:
            return res
        for i in reversed(range(len(res.segments))):
            if last == 0:
                break
            if res.segments[i].duration < min_dur:
                if i == last:
                    res.add_segments(i - 1, i, inplace=True, reassign_ids=False)
                elif i == 0:
                    res.add_segments(i, i + 1, inplace=True, reassign_ids=False)
                elif res.segments[i + 1].duration < res.segments[i - 1].duration:
                    res.add_segments(i - 1, i, inplace=True, reassign_ids=False)
                else:
                    res.add_segments(i, i + 1, inplace=True, reassign_ids=False)
                last -= 1
        res.reassign_ids()
        for sg in res.segments:
            sg.apply_min_dur(min_dur, inplace=True)
        return res

    def adjust_by_silence(self, audio, vad=False, *, verbose: Optional[bool] = False, sample_rate: Optional[int] = None,
                          vad_onnx: bool = False, vad_threshold: float = 0.35, q_levels: int = 20, k_size: int = 5,
                          min_word_dur: Optional[float] = None, min_silence_dur: Optional[float] = None,
                          word_level: bool = True, nonspeech_error: float = 0.3,
                          use_word_position: bool = True) -> "WhisperResult":
        """Silence detection on ``audio`` (the loudness-based detector) + :meth:`suppress_silence` with its timings
        (result.py:1191-1285).  ``audio``: waveform at ``sample_rate`` (16 kHz when None), path or file bytes."""
        if vad is not False:
            raise NotImplementedError("vad needs the Silero model (torch.hub, network) -- out of scope offline")
        from .audio import SAMPLE_RATE
        from .audio_io import audio_to_tensor_resample
        from .stabilization import mask2timing, wav2mask
        wave = audio_to_tensor_resample(audio, sample_rate, SAMPLE_RATE)
        timings = mask2timing(wav2mask(wave, q_levels=q_levels, k_size=k_size))
        if timings is None:
            return self
        if min_silence_dur:
            keep = (timings[1] - timings[0]) >= min_silence_dur
            timings = (timings[0][keep], timings[1][keep])
        self.suppress_silence(*timings, min_word_dur=min_word_dur, word_level=word_level, nonspeech_error=nonspeech_error,
                              use_word_position=use_word_position)
        self.update_nonspeech_sections(*timings)
        return self

    def adjust_by_result(self, other_result: "WhisperResult", min_word_dur: Optional[float] = None, verbose: bool = False):
        """Each word is narrowed to its overlap with the same word of ``other_result`` when at least ``min_word_dur``
        remains (result.py:1287-1325)."""
        if not (self.has_words and other_result.has_words):
            raise NotImplementedError("This operation can only be performed on results with word timestamps")
        mine, theirs = self.all_words(), other_result.all_words()
        assert [w.word for w in mine] == [w.word for w in theirs], "The words in [other_result] do not match the current words."
        mwd = 0.1 if min_word_dur is None else min_word_dur
        for w, o in zip(mine, theirs):
            if w.end <= o.start:
                continue
            lo, hi = max(w.start, o.start), min(w.end, o.end)
            if hi - lo < mwd:
                continue
            note = ""
            if w.start != lo:
                note += f"[Start:{w.start:.3f}->{lo:.3f}] "
                w.start = lo
            if w.end != hi:
                note += f"[End:{w.end:.3f}->{hi:.3f}]  "
                w.end = hi
            if note and verbose:
                print(f'{note}"{w.word}"')

    # -- boundary pickers of the merge operations (result.py:1341-1379) and lock groups
    def get_locked_indices(self) -> List[int]:
        from .regroup import _locked_joins
        return sorted(_locked_joins(self))

    def get_gaps(self, as_ndarray: bool = False):
        import numpy as np
        gaps = np.array([sg.start for sg in self.segments])[1:] - np.array([sg.end for sg in self.segments])[:-1]
        return gaps if as_ndarray else gaps.tolist()

    def get_gap_indices(self, min_gap: float = 0.1) -> List[int]:
        from .regroup import gap_joins
        return gap_joins(self, min_gap)

    def get_punctuation_indices(self, punctuation) -> List[int]:
        from .regroup import punctuation_joins
        return punctuation_joins(self, punctuation)

    def all_words_by_lock(self, only_text: bool = True, by_segment: bool = False, include_single: bool = False) -> list:
        if by_segment:
            return [sg.words_by_lock(only_text=only_text, include_single=include_single) for sg in self.segments]
        return group_by_lock(self.all_words(), only_text=only_text, include_single=include_single)

    def segments_to_dicts(self) -> List[dict]:
        return [sg.to_dict() for sg in self.segments]

    def find(self, pattern: str, word_level: bool = True, flags=None) -> "WhisperResultMatches":
        """Regular-expression search over the text; the matches carry the segments / words they span (result.py:3026)."""
        return WhisperResultMatches(self).find(pattern, word_level=word_level, flags=flags)

    def show_regroup_history(self):
        if not self._regroup_history:
            print("Result has no history.")
        for *_, msg in self.parse_regroup_algo(self._regroup_history):
            print(f".{msg}")

    def add_segments(self, index0: int, index1: int, inplace: bool = False, lock: bool = False, newline: bool = False,
                     reassign_ids: bool = True) -> Segment:
        """result.py:1074-1100: the two segments fused into one (words concatenated, statistics averaged)."""
        from . import regroup as R
        fused = R._fuse(self.segments[index0], self.segments[index1], newline)
        if reassign_ids:
            fused.reassign_ids()
        if lock and self.segments[index0].has_words:
            k = len(self.segments[index0].words)
            fused.words[k - 1].lock_right()
            if k < len(fused.words):
                fused.words[k].lock_left()
        if inplace:
            i0, i1 = sorted([index0, index1])
            self.segments[i0] = fused
            del self.segments[i1]
            if reassign_ids:
                self.reassign_ids(True)
        return fused

    def split_segment_by_index(self, segment: Union[int, Segment], indices: Union[int, List[int]], reassign_ids: bool = True):
        """result.py:1411-1432: cut one segment after the given word indices."""
        from . import regroup as R
        if not self.has_words:
            return
        if isinstance(indices, int):
            indices = [indices]
        elif not indices:
            return
        si = segment if isinstance(segment, int) else segment.id
        seg = self.segments[si]
        bad = [i for i in indices if i < 0 or i > len(seg.words)]
        if bad:
            raise IndexError(f"got out of split range indices: {bad}")
        parts = R._pieces(seg, list(indices))
        if reassign_ids:
            for p in parts:
                p.reassign_ids()
        self.segments[si:si + 1] = parts
        if reassign_ids:
            self.reassign_ids(True)

    def get_content_by_time(self, time, within: bool = False, segment_level: bool = False):
        """result.py:1540-1583: words (or segments) overlapping / lying within a time or (start, end) range."""
        if not segment_level and not self.has_words:
            raise ValueError("Missing word timestamps in result. Use ``segment_level=True`` instead.")
        parts = self.segments if segment_level else self.all_words()
        if isinstance(time, (float, int)):
            time = [time, time]
        elif isinstance(time, dict):
            time = [time["start"], time["end"]]
        a, b = time
        if within:
            return [c for c in parts if a <= c.start and b >= c.end]
        return [c for c in parts if a <= c.end and b >= c.start]

    # -- objects referenced from the regroup history ("<key>" place-holders, result.py:44-71)
    def _store_content(self, content) -> str:
        if content is None:
            return ""
        if isinstance(content, str):
            return content
        key = repr(content).replace("_", "-")
        if not key.startswith("<") and not key.endswith(">"):
            key = f"<{key}>"
        if not hasattr(self, "_content_cache"):
            self._content_cache = {}
        self._content_cache[key] = content
        return key

    def _get_content(self, content, strict: bool = True):
        if isinstance(content, str) and content.startswith("<") and content.endswith(">"):
            found = {"<True>": True, "<False>": False}.get(content)
            if found is None and hasattr(self, "_content_cache"):
                found = self._content_cache.get(content)
            if found is None:
                if strict:
                    raise NameError(f'{content.replace("-", "_")} not found')
                return content
            return found
        return content

    def remove_repetition(self, max_words: int = 1, case_sensitive: bool = False, strip: bool = True,
                          ignore_punctuations: str = "\"',.?!", extend_duration: bool = True, verbose: bool = True):
        from . import regroup as R
        if not self.has_words:
            return self
        R.remove_repetition(self, max_words, case_sensitive, strip, ignore_punctuations, extend_duration)
        self._log(f"rp={max_words}+{int(case_sensitive)}+{int(strip)}+{ignore_punctuations}+{int(extend_duration)}+{int(verbose)}")
        return self

    def remove_words_by_str(self, words, case_sensitive: bool = False, strip: bool = True,
                            ignore_punctuations: str = "\"',.?!", min_prob: float = None, filters=None,
                            verbose: bool = True):
        from . import regroup as R
        if not self.has_words:
            return self
        if isinstance(words, str):
            words = [words]
        elif words == 0:
            words = None
        filters = self._get_content(filters)
        R.remove_words_by_str(self, words, case_sensitive, strip, ignore_punctuations, min_prob, filters)
        shown = 0 if words is None else "/".join(R._norm_words(list(words), strip, ignore_punctuations, case_sensitive))
        self._log(f"rws={shown}+{int(case_sensitive)}+{int(strip)}+{ignore_punctuations}+{min_prob}"
                  f"+{self._store_content(filters)}+{int(verbose)}")
        return self

    def fill_in_gaps(self, other_result, min_gap: float = 0.1, case_sensitive: bool = False, strip: bool = True,
                     ignore_punctuations: str = "\"',.?!", verbose: bool = True):
        from . import regroup as R
        if len(self.segments) < 2:
            return self
        other_result = self._get_content(other_result)
        if isinstance(other_result, str):
            path, other_result = other_result, WhisperResult(other_result)
        else:
            path = self._store_content(other_result)
        R.fill_in_gaps(self, other_result, min_gap, case_sensitive, strip, ignore_punctuations)
        self._log(f"fg={path}+{min_gap}+{int(case_sensitive)}+{int(strip)}+{ignore_punctuations}+{int(verbose)}")
        return self

    def adjust_gaps(self, duration_threshold: float = 0.75, one_section: bool = False):
        from . import regroup as R
        R.adjust_gaps(self, duration_threshold, one_section)
        self._log(f"ag={duration_threshold}+{int(one_section)}")
        return self

    def custom_operation(self, key: str, operator, value, method, word_level: Optional[bool] = None):
        from . import regroup as R
        parts = R.custom_operation(self, key, operator, value, method, word_level)
        self._log("co=" + "+".join(str(x) for x in parts))
        return self

    def to_dict(self, keep_orig: bool = True) -> dict:
        ori = self.ori_dict if keep_orig else {}
        return dict(text=self.text, segments=[s.to_dict() for s in self.segments], language=self.language,
                    ori_dict=ori, regroup_history=self._regroup_history,
                    nonspeech_sections=self._nonspeech_sections, unfinished=self.unfinished_start)

    def save_as_json(self, path: str):
        d = self.to_dict(keep_orig=False)
        d.pop("ori_dict", None)
        with open(path, "w", encoding="utf-8") as f:
            json.dump(d, f, allow_nan=True)

    def suppress_silence(self, silent_starts, silent_ends, min_word_dur: Optional[float] = None, word_level: bool = True,
                         nonspeech_error: float = 0.3, use_word_position: bool = True, verbose: bool = False):
        """result.py:1137-1189: move timestamps lying in the given non-speech sections to the sections' boundaries."""
        import numpy as np
        s, e = np.asarray(silent_starts), np.asarray(silent_ends)
        for seg in self.segments:
            seg.suppress_silence(s, e, min_word_dur, word_level=word_level, nonspeech_error=nonspeech_error,
                                 use_word_position=use_word_position)
        return self

    def set_current_as_orig(self, keep_orig: bool = False):
        """result.py:3076-3080: the current state becomes what ``reset()`` returns to."""
        self.ori_dict = self.to_dict(keep_orig=keep_orig)

    def reset(self):
        """result.py:3082-3092: back to the segments this result was created with."""
        self.language = self.ori_dict.get("language")
        self._regroup_history = ""
        self.segments = [Segment(**s) for s in (self.ori_dict.get("segments") or [])]
        if self._forced_order:
            self.force_order()
        self.remove_no_word_segments(any(s.has_words for s in self.segments))

    # -- regrouping entry points (bodies in regroup.py)
    def _log(self, entry: str):
        if entry:
            self._regroup_history += ("_" if self._regroup_history else "") + entry

    def ignore_special_periods(self, enable: bool = True):
        self._ignore_special_periods = enable
        self._log(f"isp={int(enable)}")
        return self

    def split_by_gap(self, max_gap: float = 0.1, lock: bool = False, newline: bool = False,
                     ignore_special_periods: bool = False):
        from . import regroup as R
        isp = self._ignore_special_periods or ignore_special_periods
        R.split_segments(self, lambda s: R.gap_cuts(s, max_gap), lock=lock, newline=newline, skip_special_periods=isp)
        self._log(f"sg={max_gap}+{int(lock)}+{int(newline)}+{int(isp)}")
        return self

    def split_by_punctuation(self, punctuation, lock: bool = False, newline: bool = False,
                             min_words: Optional[int] = None, min_chars: Optional[int] = None,
                             min_dur: Optional[float] = None, ignore_special_periods: bool = False):
        from . import regroup as R
        eligible = None
        if any((min_words, min_chars, min_dur)):
            eligible = {id(s) for s in self.segments
                        if (min_words and len(s.words or ()) >= min_words) or (min_chars and s.char_count() >= min_chars)
                        or (min_dur and s.duration >= min_dur)}
        isp = self._ignore_special_periods or ignore_special_periods
        R.split_segments(self, lambda s: R.punctuation_cuts(s, punctuation)
                         if eligible is None or id(s) in eligible else [],
                         lock=lock, newline=newline, skip_special_periods=isp)
        self._log(f"sp={R.punctuation_str(punctuation)}+{int(lock)}+{int(newline)}+{min_words or ''}"
                  f"+{min_chars or ''}+{min_dur or ''}+{int(isp)}")
        return self

    def split_by_length(self, max_chars: int = None, max_words: int = None, even_split: bool = True,
                        force_len: bool = False, lock: bool = False, include_lock: bool = False,
                        newline: bool = False, ignore_special_periods: bool = False):
        from . import regroup as R
        if force_len:
            self.merge_all_segments(record=False)
        isp = self._ignore_special_periods or ignore_special_periods
        R.split_segments(self, lambda s: R.length_cuts(s, max_chars, max_words, even_split, include_lock),
                         lock=lock, newline=newline, skip_special_periods=isp)
        self._log(f"sl={max_chars or ''}+{max_words or ''}+{int(even_split)}+{int(force_len)}"
                  f"+{int(lock)}+{int(include_lock)}+{int(newline)}+{int(isp)}")
        return self

    def split_by_duration(self, max_dur: float, even_split: bool = True, force_len: bool = False, lock: bool = False,
                          include_lock: bool = False, newline: bool = False, ignore_special_periods: bool = False):
        from . import regroup as R
        if force_len:
            self.merge_all_segments(record=False)
        isp = self._ignore_special_periods or ignore_special_periods
        R.split_segments(self, lambda s: R.duration_cuts(s, max_dur, even_split, include_lock),
                         lock=lock, newline=newline, skip_special_periods=isp)
        self._log(f"sd={max_dur}+{int(even_split)}+{int(force_len)}+{int(lock)}+{int(include_lock)}+{int(newline)}"
                  f"+{int(isp)}")
        return self

    def merge_by_gap(self, min_gap: float = 0.1, max_words: int = None, max_chars: int = None,
                     is_sum_max: bool = False, lock: bool = False, newline: bool = False):
        from . import regroup as R
        R.merge_segments(self, R.gap_joins(self, min_gap), max_words=max_words, max_chars=max_chars,
                         is_sum_max=is_sum_max, lock=lock, newline=newline)
        self._log(f"mg={min_gap}+{max_words or ''}+{max_chars or ''}+{int(is_sum_max)}+{int(lock)}+{int(newline)}")
        return self

    def merge_by_punctuation(self, punctuation, max_words: int = None, max_chars: int = None,
                             is_sum_max: bool = False, lock: bool = False, newline: bool = False):
        from . import regroup as R
        R.merge_segments(self, R.punctuation_joins(self, punctuation), max_words=max_words, max_chars=max_chars,
                         is_sum_max=is_sum_max, lock=lock, newline=newline)
        self._log(f"mp={R.punctuation_str(punctuation)}+{max_words or ''}+{max_chars or ''}+{int(is_sum_max)}"
                  f"+{int(lock)}+{int(newline)}")
        return self

    def merge_all_segments(self, record: bool = True):
        from . import regroup as R
        R.merge_all(self)
        if record and self.segments:
            self._log("ms")
        return self

    def clamp_max(self, medium_factor: float = 2.5, max_dur: float = None, clip_start: Optional[bool] = None,
                  verbose: bool = False):
        from . import regroup as R
        if not (medium_factor or max_dur):
            raise ValueError("At least one of following arguments requires non-zero value: medium_factor; max_dur")
        if not self.has_words:
            warnings.warn("Cannot clamp due to missing/no word-timestamps")
            return self
        R.clamp_word_durations(self, medium_factor, max_dur, clip_start)
        self._log(f"cm={medium_factor}+{max_dur or ''}+{clip_start or ''}+{int(verbose)}")
        return self

    def lock(self, startswith=None, endswith=None, right: bool = True, left: bool = False,
             case_sensitive: bool = False, strip: bool = True):
        from . import regroup as R
        assert startswith is not None or endswith is not None, "Must specify [startswith] or/and [endswith]."
        pre, suf = R.lock_matching(self, startswith, endswith, right, left, case_sensitive, strip)
        self._log(f"l={'/'.join(pre)}+{'/'.join(suf)}+{int(right)}+{int(left)}+{int(case_sensitive)}+{int(strip)}")
        return self

    def unlock_all_segments(self):
        for s in self.segments:
            s.unlock_all_words()
        return self

    def pad(self, start_pad: Optional[float] = None, end_pad: Optional[float] = None, max_dur: Optional[float] = None,
            max_end: Optional[float] = None, word_level: bool = False):
        from . import regroup as R
        if not (start_pad or end_pad):
            warnings.warn("No ``start_pad`` or ``end_pad`` given.", stacklevel=2)
            return self
        word_level = bool(word_level and self.has_words)
        R.pad_parts(self.all_words() if word_level else self.segments, start_pad, end_pad, max_dur, max_end)
        self._log(f"p={start_pad or ''}+{end_pad or ''}+{max_dur or ''}+{max_end or ''}+{int(word_level)}")
        return self

    def convert_to_segment_level(self):
        for s in self.segments:
            s.convert_to_segment_level()
        self._log("csl")
        return self

    def remove_word(self, word, reassign_ids: bool = True, verbose: bool = True, record: bool = True):
        """result.py:2149-2194.  ``word`` is a WordTiming, a (segment, word) index pair or "seg,word"."""
        if isinstance(word, WordTiming):
            if self[word.segment_id][word.id] is not word:
                self.reassign_ids()
                if self[word.segment_id][word.id] is not word:
                    raise ValueError("word not in result")
            si, wi = word.segment_id, word.id
        else:
            si, wi = map(int, word.split(",")) if isinstance(word, str) else word
        if verbose:
            print(f"Removed: {self[si][wi].to_dict()}")       # a segment without words raises ValueError here, as upstream
        del self.segments[si].words[wi]
        if not reassign_ids:
            return self
        if self.segments[si].has_words:
            self.segments[si].reassign_ids()
        else:
            self.remove_no_word_segments()
        if record:
            self._log(f"rw={si},{wi}+{int(reassign_ids)}+{int(verbose)}")
        return self

    def remove_segment(self, segment, reassign_ids: bool = True, verbose: bool = True, record: bool = True):
        """result.py:2196-2236."""
        if isinstance(segment, Segment):
            if self.segments[segment.id] is not segment:
                self.reassign_ids()
                if self.segments[segment.id] is not segment:
                    raise ValueError("segment not in result")
            segment = segment.id
        if verbose:
            print(f"Removed: [id:{segment}] {self.segments[segment]!r}")
        del self.segments[segment]
        if not reassign_ids:
            return self
        self.reassign_ids(True, start=segment)
        if record:
            self._log(f"rs={segment}+{int(reassign_ids)}+{int(verbose)}")
        return self

    def regroup(self, regroup_algo: Union[str, bool, None] = None, verbose: bool = False, only_show: bool = False):
        """result.py:2893-2978: run a regrouping program (``True``/``None`` = the default algorithm 'da')."""
        from . import regroup as R
        if regroup_algo is False:
            return self
        if regroup_algo is None or regroup_algo is True:
            regroup_algo = "da"
        for method, kwargs, shown in R.parse_regroup_algo(self, regroup_algo, include_str=verbose or only_show):
            if shown:
                print(shown)
            if not only_show:
                method(**kwargs)
        return self

    def parse_regroup_algo(self, regroup_algo: str, include_str: bool = True):
        from . import regroup as R
        return R.parse_regroup_algo(self, regroup_algo, include_str)


class SegmentMatch:
    """One regular-expression match: the segments it touches and, for a word-level search, the words (result.py:3105)."""

    def __init__(self, segments: Union[List[Segment], Segment], _word_indices: Optional[List[List[int]]] = None,
                 _text_match: Optional[str] = None):
        self.segments = [segments] if isinstance(segments, Segment) else segments
        self.word_indices = [] if _word_indices is None else _word_indices
        self.words = [self.segments[i].words[j] for i, idx in enumerate(self.word_indices) for j in idx]
        self.text = "".join(w.word for w in self.words) if self.words else "".join(sg.text for sg in self.segments)
        self.text_match = _text_match

    @property
    def start(self):
        if self.words:
            return self.words[0].start
        return self.segments[0].start if self.segments else None

    @property
    def end(self):
        if self.words:
            return self.words[-1].end
        return self.segments[-1].end if self.segments else None

    def __len__(self):
        return len(self.segments)

    def __repr__(self):
        return repr(self.__dict__)

    __str__ = __repr__


class WhisperResultMatches:
    """Matches of ``WhisperResult.find``; ``find`` on it searches again inside the matched segments (result.py:3152)."""

    def __init__(self, matches: Union[List[SegmentMatch], WhisperResult], _segment_indices: Optional[List[List[int]]] = None):
        if isinstance(matches, WhisperResult):
            self.matches = [SegmentMatch(sg) for sg in matches.segments]
            self._segment_indices = [[i] for i in range(len(matches.segments))]
        else:
            assert _segment_indices is not None and len(matches) == len(_segment_indices)
            assert all(len(m.segments) == len(_segment_indices[i]) for i, m in enumerate(matches))
            self.matches, self._segment_indices = matches, _segment_indices

    @property
    def segment_indices(self) -> List[List[int]]:
        return self._segment_indices

    def _runs(self) -> List[List[tuple]]:
        """the distinct matched segments as (index, segment) in order, cut into runs of consecutive indices -- with the
        reference's cut placement (a run is closed when the segment just added does NOT follow its predecessor)"""
        runs, cur, top = [], [], -1
        for idx, match in zip(self._segment_indices, self.matches):
            for i, sg in zip(sorted(idx), match.segments):
                if i > top:
                    cur.append((i, sg))
                    if i - 1 != top:
                        runs.append(cur)
                        cur = []
                    top = i
        if cur:
            runs.append(cur)
        return runs

    def find(self, pattern: str, word_level: bool = True, flags=None) -> "WhisperResultMatches":
        import re
        if word_level and not all(sg.has_words for m in self.matches for sg in m.segments):
            warnings.warn("Cannot perform word-level search with segment(s) missing word timestamps.")
            word_level = False
        found, found_idx = [], []
        for run in self._runs():
            if word_level:
                owner = [(i, j) for i, sg in run for j, w in enumerate(sg.words) for _ in w.word]
                text = "".join(w.word for _, sg in run for w in sg.words)
            else:
                owner = [(i, None) for i, sg in run for _ in sg.text]
                text = "".join(sg.text for _, sg in run)
            for m in re.finditer(pattern, text, flags=flags or 0):
                span = owner[m.start(): m.end()]
                seg_ids = sorted({i for i, _ in span})
                words = [sorted({j for i, j in span if i == k}) for k in seg_ids] if word_level else None
                found.append(SegmentMatch([sg for i, sg in run if i in seg_ids], _word_indices=words, _text_match=m.group()))
                found_idx.append(seg_ids)
        return WhisperResultMatches(found, found_idx)

    def __len__(self):
        return len(self.matches)

    def __bool__(self):
        return len(self.matches) != 0

    def __getitem__(self, i) -> SegmentMatch:
        return self.matches[i]
