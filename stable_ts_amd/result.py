"""Result containers returned by transcribe()/align().

The reference's result model (stable_whisper/result.py: WordTiming :74-257, Segment :277-925, WhisperResult
:928-3102) is pure-Python post-processing that SURVEY.md 8f ranks as "next-1" (regroup DSL, silence snapping, splitting
and merging).  This module carries the part the hot path produces and consumers read: the same field names and the
same ``to_dict()`` JSON schema (result.py:1398-1406), ordering checks, and word/segment accessors.
"""
import json
import warnings
from typing import Iterator, List, Optional, Union


class WordTiming:
    def __init__(self, word: str, start: float, end: float, probability: Optional[float] = None,
                 tokens: Optional[List[int]] = None, segment_id: Optional[int] = None, id: Optional[int] = None, **_):
        self.word = word
        self.start = float(start)
        self.end = float(end)
        self.probability = probability
        self.tokens = tokens
        self.segment_id = segment_id
        self.id = id

    @property
    def duration(self) -> float:
        return round(self.end - self.start, 3)

    def offset_time(self, offset: float):
        self.start = round(self.start + offset, 3)
        self.end = round(self.end + offset, 3)

    def to_dict(self) -> dict:
        return dict(word=self.word, start=self.start, end=self.end, probability=self.probability, tokens=self.tokens,
                    segment_id=self.segment_id, id=self.id)

    def __repr__(self):
        return f"WordTiming({self.word!r}, {self.start}, {self.end})"


class Segment:
    def __init__(self, start: Optional[float] = None, end: Optional[float] = None, text: Optional[str] = None,
                 seek: Optional[float] = None, tokens: Optional[List[int]] = None, temperature: Optional[float] = None,
                 avg_logprob: Optional[float] = None, compression_ratio: Optional[float] = None,
                 no_speech_prob: Optional[float] = None, words: Optional[List[Union[WordTiming, dict]]] = None,
                 id: Optional[int] = None, **_):
        self._start, self._end, self._text = start, end, text
        self.seek = seek
        self.tokens = tokens
        self.temperature = temperature
        self.avg_logprob = avg_logprob
        self.compression_ratio = compression_ratio
        self.no_speech_prob = no_speech_prob
        self.id = id
        self.words: Optional[List[WordTiming]] = None
        if words is not None:
            self.words = [w if isinstance(w, WordTiming) else WordTiming(**w) for w in words]
            for i, w in enumerate(self.words):
                w.segment_id, w.id = id, i

    @property
    def has_words(self) -> bool:
        return bool(self.words)

    @property
    def start(self) -> float:
        return self.words[0].start if self.has_words else self._start

    @property
    def end(self) -> float:
        return self.words[-1].end if self.has_words else self._end

    @property
    def text(self) -> str:
        return "".join(w.word for w in self.words) if self.has_words else self._text

    @property
    def duration(self) -> float:
        return self.end - self.start

    def offset_time(self, offset: float):
        if self.seek is not None:
            self.seek = round(self.seek + offset, 3)
        if self.has_words:
            for w in self.words:
                w.offset_time(offset)
        else:
            self._start = round(self._start + offset, 3)
            self._end = round(self._end + offset, 3)

    def to_dict(self) -> dict:
        d = dict(start=self.start, end=self.end, text=self.text, seek=self.seek, tokens=self.tokens,
                 temperature=self.temperature, avg_logprob=self.avg_logprob, compression_ratio=self.compression_ratio,
                 no_speech_prob=self.no_speech_prob, id=self.id)
        d["words"] = [w.to_dict() for w in self.words] if self.words is not None else None
        return d

    def __repr__(self):
        return f"Segment({self.start}, {self.end}, {self.text!r})"


class UnsortedException(Exception):
    pass


class WhisperResult:
    def __init__(self, result: Union[dict, list, str], force_order: bool = False, check_sorted: bool = True):
        if isinstance(result, str):
            with open(result, "r", encoding="utf-8") as f:
                result = json.load(f)
        if isinstance(result, list):
            result = dict(segments=result)
        self.ori_dict = result.get("ori_dict") or result
        self.language = result.get("language")
        self.unfinished_start = result.get("unfinished_start", -1.0)
        self.nonspeech_sections = result.get("nonspeech_sections", [])
        segs = result.get("segments") or []
        self.segments: List[Segment] = [s if isinstance(s, Segment) else Segment(**s) for s in segs]
        self._text = result.get("text")
        self.reassign_ids()
        if force_order:
            self.force_order()
        if check_sorted:
            self.raise_for_unsorted()
        self.remove_no_word_segments()

    # -- container protocol
    def __len__(self):
        return len(self.segments)

    def __getitem__(self, i):
        return self.segments[i]

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

    def all_words(self) -> List[WordTiming]:
        return [w for s in self.segments for w in (s.words or [])]

    def all_tokens(self) -> List[int]:
        return [t for s in self.segments for t in (s.tokens or [])]

    def reassign_ids(self):
        for i, s in enumerate(self.segments):
            s.id = i
            for j, w in enumerate(s.words or []):
                w.segment_id, w.id = i, j

    def remove_no_word_segments(self):
        """result.py:948: segments that were given a (now empty) word list are dropped."""
        self.segments = [s for s in self.segments if s.words is None or len(s.words) > 0]
        self.reassign_ids()

    def force_order(self):
        prev_end = 0.0
        for s in self.segments:
            if s.has_words:
                continue
            if s._start < prev_end:
                s._start = prev_end
            if s._end < s._start:
                s._end = s._start
            prev_end = s._end

    def raise_for_unsorted(self):
        """result.py:1020-1056: every timestamp must be non-decreasing."""
        stamps = []
        for s in self.segments:
            if s.has_words:
                for w in s.words:
                    stamps.extend([w.start, w.end])
            else:
                stamps.extend([s.start, s.end])
        for a, b in zip(stamps[:-1], stamps[1:]):
            if b < a:
                raise UnsortedException(f"timestamps are not in ascending order: {a} -> {b}")

    def offset_time(self, offset: float):
        for s in self.segments:
            s.offset_time(offset)

    def add_segments(self, other: "WhisperResult"):
        self.segments.extend(other.segments)
        self.reassign_ids()

    def to_dict(self) -> dict:
        return dict(text=self.text, segments=[s.to_dict() for s in self.segments], language=self.language,
                    ori_dict=self.ori_dict if isinstance(self.ori_dict, dict) and self.ori_dict is not self.__dict__ else None,
                    nonspeech_sections=self.nonspeech_sections, unfinished_start=self.unfinished_start)

    def save_as_json(self, path: str):
        d = self.to_dict()
        d.pop("ori_dict", None)
        with open(path, "w", encoding="utf-8") as f:
            json.dump(d, f, allow_nan=True)

    def regroup(self, *_, **__):
        warnings.warn("regroup() is post-processing outside this round's hot-path scope (SURVEY.md 8f next-1); "
                      "segments are returned as decoded")
        return self
