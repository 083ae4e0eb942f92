"""Regrouping of word-timed segments: the algorithms behind ``WhisperResult.split_by_* / merge_by_* / clamp_max /
regroup`` and the regroup string DSL.

Behavioural contract = stable_whisper/result.py (index pickers :707-884, ``Segment.split`` :886-902,
``_split_segments`` :1446-1494, ``_merge_segments`` :1496-1531, ``clamp_max`` :2022-2080, ``lock`` :2082-2147,
``pad`` :1798-1861, ``merge_all_segments`` :1863-1894, ``regroup``/``parse_regroup_algo`` :2893-3024).  The default
program applied by transcribe()/align() is ``isp_cm_sp=.* /。/?/？_sg=.5_sp=,* /，++++50_sl=70_cm`` (:3008).

Organisation here: a *cut* is the index of the last word of a would-be segment ("cut after word i").  Each ``*_cuts``
function maps one segment to its cut list (vectorised over the words with numpy where that is natural), each
``*_joins`` function maps the result to the list of segment boundaries to dissolve, and ``split_segments`` /
``merge_segments`` apply them.  Host-only; parity is checked word-for-word against the reference's own
``WhisperResult`` in tests/test_regroup_cpu.py (live when /root/reference is importable, golden fixtures otherwise).
"""
import re
from itertools import chain
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np

from .result import Segment, WhisperResult, _blend

DEFAULT_ALGO = "isp_cm_sp=.* /。/?/？_sg=.5_sp=,* /，++++50_sl=70_cm"

Punct = Union[str, Sequence[Union[str, Sequence[str]]]]


# --------------------------------------------------------------------------------------------------------- pickers
def _locked_cuts(words) -> List[int]:
    """Boundaries that an earlier ``lock=True`` operation pinned (result.py:707-711)."""
    return [i for i in range(len(words) - 1) if words[i + 1].left_locked or words[i].right_locked]


def gap_cuts(seg: Segment, max_gap: Optional[float]) -> List[int]:
    ws = seg.words
    if not ws or len(ws) < 2:
        return []
    gaps = np.array([w.start for w in ws[1:]]) - np.array([w.end for w in ws[:-1]])
    hit = set(np.flatnonzero(gaps > (max_gap or 0)).tolist())
    return sorted(hit - set(_locked_cuts(ws)))


def _punct_hits(texts: List[str], punctuation: Punct) -> set:
    """Shared by word-level splitting (:729-747) and segment-level merging (:1361-1379): boundary i sits between
    texts[i] and texts[i+1].  A plain string matches as a suffix of the left item or (except at the very first item)
    as a prefix of the left item, which moves the boundary one to the left; a pair matches (suffix, next prefix)."""
    if isinstance(punctuation, str):
        punctuation = [punctuation]
    hits = set()
    n = len(texts)
    for p in punctuation:
        if isinstance(p, str):
            for i in range(n - 1):
                if texts[i].endswith(p):
                    hits.add(i)
                elif i and texts[i].startswith(p):
                    hits.add(i - 1)
        else:
            tail, head = p
            hits.update(i for i in range(n - 1) if texts[i].endswith(tail) and texts[i + 1].startswith(head))
    return hits


def punctuation_cuts(seg: Segment, punctuation: Punct) -> List[int]:
    ws = seg.words
    if not ws or len(ws) < 2:
        return []
    return sorted(_punct_hits([w.word for w in ws], punctuation) - set(_locked_cuts(ws)))


_CAP_OR_DIGIT = re.compile(r"^[A-Z0-9]")
_NOT_ABBREV = re.compile(r"[.A-Z0-9]")


def _is_abbreviation(word: str) -> bool:
    """'Mr.', ' U.S.', ' 3.' ...: starts with a capital/digit and has fewer than 3 other characters (:1434-1444)."""
    return _CAP_OR_DIGIT.search(word) is not None and len(_NOT_ABBREV.sub("", word)) < 3


def _special_period_words(seg: Segment, extra: Optional[List[int]] = None) -> List[int]:
    """Segment._get_special_period_indices (:749-758); note: tests the raw (unstripped) word, like the reference."""
    idx = [i for i, w in enumerate(seg.words)
           if _CAP_OR_DIGIT.search(w.word) is not None and not w.word.endswith("..") and
           len(_NOT_ABBREV.sub("", w.word)) < 3]
    return sorted(set(idx + extra)) if extra else idx


def _nearest_targets(cum: np.ndarray, per_part: float, parts: float, avoid: Optional[List[int]], n_words: int):
    """Even splitting (:760-786): for k = 1..parts-1 choose the word whose running total is nearest k*per_part.
    Word positions in ``avoid`` are aliased to their right neighbour so a cut never lands on them."""
    cum = np.asarray(cum, dtype=np.float64).copy()
    targets = [k * per_part for k in range(1, int(parts))]
    if avoid:
        alias = np.arange(len(cum))
        for i in sorted(set(avoid)):
            if i == n_words - 1:
                break
            cum[i] = cum[i + 1]
            alias[i] = alias[i + 1]
        return sorted({int(alias[int(np.abs(cum - t).argmin())]) for t in targets})
    return [int(np.abs(cum - t).argmin()) for t in targets]


def length_cuts(seg: Segment, max_chars: Optional[int] = None, max_words: Optional[int] = None,
                even_split: bool = True, include_lock: bool = False, ignore_special_periods: bool = False) -> List[int]:
    ws = seg.words
    if not ws or (max_chars is None and max_words is None):
        return []
    assert max_chars != 0 and max_words != 0, \
        f"max_chars and max_words must be greater 0, but got {max_chars} and {max_words}"
    n = len(ws)
    if n < 2:
        return []
    avoid = _locked_cuts(ws) if include_lock else []
    if ignore_special_periods:
        avoid = _special_period_words(seg, avoid)
    lens = [len(w.word) for w in ws]
    if not even_split:                                      # greedy fill (:838-851)
        cuts, n_w, n_c = [], 0, 0
        for i in range(n):
            n_w += 1
            n_c += lens[i]
            over = (max_chars is not None and n_c > max_chars) or (max_words is not None and n_w > max_words)
            if i and over and (i - 1) not in avoid:
                cuts.append(i - 1)
                n_w, n_c = 1, lens[i]
        return cuts
    cuts: List[int] = []
    too_many_words = max_words is not None and n > max_words
    if max_chars is not None and sum(lens) > max_chars:
        total = sum(lens)
        parts = np.ceil(total / max_chars)
        cuts = _nearest_targets(np.cumsum(lens[:-1]), total / parts, parts, avoid, n)
        if max_words is not None:
            too_many_words = any(b - a + 1 > max_words for a, b in zip([0] + cuts, cuts + [n]))
    if too_many_words:
        parts = np.ceil(n / max_words)
        cuts = _nearest_targets(np.arange(1, n + 1), n / parts, parts, avoid, n)
    return cuts


def duration_cuts(seg: Segment, max_dur: float, even_split: bool = True, include_lock: bool = False,
                  ignore_special_periods: bool = False) -> List[int]:
    ws = seg.words
    if not ws:
        return []
    durs = [w.duration for w in ws]
    total = np.sum(durs)
    if total <= max_dur:
        return []
    avoid = _locked_cuts(ws) if include_lock else []
    if ignore_special_periods:
        avoid = _special_period_words(seg, avoid)
    if even_split:
        parts = np.ceil(total / max_dur)
        return _nearest_targets(np.cumsum(durs[:-1]), total / parts, parts, avoid, len(ws))
    cuts, acc = [], 0.0
    for i, d in enumerate(durs):
        acc += d
        if i and acc > max_dur and (i - 1) not in avoid:
            cuts.append(i - 1)
            acc = d
    return cuts


# ------------------------------------------------------------------------------------------------------- splitting
def _pieces(seg: Segment, cuts: List[int]) -> List[Segment]:
    """Segment.split (:886-902): consecutive word runs ending at each cut (the tail run is implicit)."""
    n = len(seg.words)
    edges = [c + 1 for c in cuts]
    if not edges or edges[-1] != n:
        edges.append(n)
    out, lo = [], 0
    for hi in edges:
        if hi > lo:
            out.append(seg.spawn(seg.words[lo:hi]))
        lo = hi
    return out


def split_segments(result: WhisperResult, pick: Callable[[Segment], List[int]], *, lock: bool = False,
                   newline: bool = False, skip_special_periods: bool = False):
    """Apply a cut picker to every segment (:1446-1494).  With ``newline`` the cut becomes a line break inside the
    word instead of a new segment; with ``lock`` the new boundaries are pinned against later splits/merges."""
    import warnings
    rebuilt: List[Segment] = []
    saw_wordless = False
    for seg in result.segments:
        saw_wordless = saw_wordless or not seg.has_words
        cuts = sorted(set(pick(seg)))
        if skip_special_periods:
            cuts = [c for c in cuts
                    if not (seg.words[c].word.endswith(".") and _is_abbreviation(seg.words[c].word.strip()))]
        if cuts and newline:
            ws = seg.words
            if cuts[-1] == len(ws) - 1:
                cuts = cuts[:-1]
            for c in cuts:
                if ws[c].word.endswith("\n"):
                    continue
                ws[c].word += "\n"
                if lock:
                    ws[c].lock_right()
                    if c + 1 < len(ws):
                        ws[c + 1].lock_left()
            cuts = []
        if not cuts:
            rebuilt.append(seg)
            continue
        parts = _pieces(seg, cuts)
        if lock:
            for k, p in enumerate(parts):
                if k == 0:
                    p.lock_right()
                elif k == len(parts) - 1:
                    p.lock_left()
                else:
                    p.lock_both()
        rebuilt.extend(parts)
    result.segments = rebuilt
    if saw_wordless:
        warnings.warn("Found segment(s) without word timings. These segment(s) cannot be split.")
    result.remove_no_word_segments()


# --------------------------------------------------------------------------------------------------------- merging
def _locked_joins(result: WhisperResult) -> set:
    s = result.segments
    return {i for i in range(len(s) - 1) if s[i + 1].left_locked or s[i].right_locked}


def gap_joins(result: WhisperResult, min_gap: Optional[float]) -> List[int]:
    s = result.segments
    if len(s) < 2:
        return []
    gaps = np.array([x.start for x in s[1:]]) - np.array([x.end for x in s[:-1]])
    return sorted(set(np.flatnonzero(gaps <= (min_gap or 0)).tolist()) - _locked_joins(result))


def punctuation_joins(result: WhisperResult, punctuation: Punct) -> List[int]:
    if len(result.segments) < 2:
        return []
    return sorted(_punct_hits([s.text for s in result.segments], punctuation) - _locked_joins(result))


def _fuse(a: Segment, b: Segment, newline: bool) -> Segment:
    """Segment.add (:466-492): words concatenated, decode statistics averaged."""
    if a.ori_has_words != b.ori_has_words:
        raise ValueError("Can't merge segment %s words and a segment %s words." %
                         ("with" if a.ori_has_words else "without", "with" if b.ori_has_words else "without"))
    out = a.spawn((a.words + b.words) if a.ori_has_words else None)
    for k in ("temperature", "avg_logprob", "compression_ratio", "no_speech_prob"):
        setattr(out, k, _blend(getattr(a, k), getattr(b, k)))
    if a.ori_has_words:
        out._default_end = b._default_end
        out._default_text, out._default_tokens = b._default_text, list(b._default_tokens)
    else:
        out._default_start, out._default_end = a._default_start, b._default_end
        out._default_text = a._default_text + b._default_text
        out._default_tokens = list(a._default_tokens) + list(b._default_tokens)
    if newline:
        if out.words:
            last = out.words[len(a.words) - 1]
            if not last.word.endswith("\n"):
                last.word += "\n"
        elif a.text and a.text[-1] != "\n":
            out._default_text = a.text + "\n" + b.text
    return out


def merge_segments(result: WhisperResult, joins: List[int], *, max_words: Optional[int] = None,
                   max_chars: Optional[int] = None, is_sum_max: bool = False, lock: bool = False,
                   newline: bool = False):
    """Dissolve the listed boundaries right-to-left unless a size limit vetoes it (:1496-1531).  Without
    ``is_sum_max`` a merge is vetoed only when BOTH neighbours already exceed the limit."""
    segs = result.segments
    for i in reversed(joins):
        a, b = segs[i], segs[i + 1]
        if max_words and a.has_words:
            wa, wb = a.word_count(), b.word_count()
            if (wa + wb > max_words) if is_sum_max else (wa > max_words and wb > max_words):
                continue
        if max_chars:
            ca, cb = a.char_count(), b.char_count()
            if (ca + cb > max_chars) if is_sum_max else (ca > max_chars and cb > max_chars):
                continue
        fused = _fuse(a, b, newline)
        if lock and a.has_words:
            k = len(a.words)
            fused.words[k - 1].lock_right()
            if k < len(fused.words):
                fused.words[k].lock_left()
        segs[i:i + 2] = [fused]
    result.remove_no_word_segments()


def merge_all(result: WhisperResult):
    segs = result.segments
    if not segs:
        return
    if result.has_words:
        one = segs[0].spawn(result.all_words())
    else:
        one = segs[0]
        one._default_text = "".join(s.text for s in segs)
        if all(s.tokens is not None for s in segs):
            one._default_tokens = list(chain.from_iterable(s.tokens for s in segs))
        one.end = segs[-1].end
    result.segments = [one]
    result.reassign_ids()


# ----------------------------------------------------------------------------------------------- timestamp edits
def clamp_word_durations(result: WhisperResult, medium_factor: Optional[float], max_dur: Optional[float],
                         clip_start: Optional[bool]):
    """Per segment, cap word durations at ``medium_factor`` x the segment's (upper) median word duration and/or
    ``max_dur``; by default only the first word (from its start) and the last word (from its end) are clipped."""
    for seg in result.segments:
        cap = None
        if medium_factor and len(seg.words) > 1:
            d = np.sort(np.array([w.duration for w in seg.words]))
            cap = medium_factor * d[len(d) // 2]
        if max_dur and (not cap or cap > max_dur):
            cap = max_dur
        if not cap:
            continue
        if clip_start is None:
            seg.words[0].clamp_max(cap, clip_start=True)
            seg.words[-1].clamp_max(cap, clip_start=False)
        else:
            for w in seg.words:
                w.clamp_max(cap, clip_start=clip_start)


def lock_matching(result: WhisperResult, startswith, endswith, right: bool, left: bool, case_sensitive: bool,
                  strip: bool) -> Tuple[List[str], List[str]]:
    def norm(xs):
        xs = [] if xs is None else ([xs] if isinstance(xs, str) else list(xs))
        if not case_sensitive:
            xs = [x.lower() for x in xs]
        return [x.strip() for x in xs] if strip else xs

    pre, suf = norm(startswith), norm(endswith)
    for part in result.all_words_or_segments():
        text = part.word if hasattr(part, "word") else part.text
        text = text if case_sensitive else text.lower()
        text = text.strip() if strip else text
        n_hit = sum(text.startswith(p) for p in pre) + sum(text.endswith(s) for s in suf)
        if n_hit:
            if right:
                part.lock_right()
            if left:
                part.lock_left()
    return pre, suf


def pad_parts(parts: list, start_pad: Optional[float], end_pad: Optional[float], max_dur: Optional[float],
              max_end: Optional[float]):
    """pad (:1798-1861): extend starts backwards / ends forwards without crossing the neighbours."""
    assert not start_pad or start_pad > 0, "``start_pad`` must be positive"
    assert not end_pad or end_pad > 0, "``end_pad`` must be positive"
    assert max_dur is None or max_dur > 0, "``max_dur`` must be greater than 0"
    assert max_end is None or max_end > 0, "``max_end`` must be greater than 0"
    for i, p in enumerate(parts):
        if max_dur and p.end - p.start > max_dur:
            continue
        if start_pad:
            p.start = max(parts[i - 1].end if i else 0, p.start - start_pad)
        if end_pad:
            limit = max_end
            if i + 1 < len(parts):
                nxt = parts[i + 1].start
                limit = min(max_end, nxt) if max_end else nxt
            new_end = p.end + end_pad
            if limit and limit < new_end:
                new_end = limit
            if new_end > p.end:
                p.end = new_end


# ------------------------------------------------------------------------------------------------------------ DSL
def punctuation_str(punctuation: Punct) -> str:
    if isinstance(punctuation, str):
        return "/".join(punctuation)         # the reference joins over the characters of a bare string (:1744)
    return "/".join(p if isinstance(p, str) else "*".join(p) for p in punctuation)


def _parse_value(v: str):
    """utils.py:20-30: '' -> None; 'a/b*c' -> ['a', ['b', 'c']]; numerals -> int/float; anything else stays a string."""
    if v == "":
        return None
    if "/" in v:
        return [a.split("*") if "*" in a else a for a in v.split("/")]
    try:
        return float(v) if "." in v else int(v)
    except ValueError:
        return v


# key -> (method name, positional parameter names)
_OPS = dict(
    sg=("split_by_gap", ("max_gap", "lock", "newline", "ignore_special_periods")),
    sp=("split_by_punctuation", ("punctuation", "lock", "newline", "min_words", "min_chars", "min_dur",
                                 "ignore_special_periods")),
    sl=("split_by_length", ("max_chars", "max_words", "even_split", "force_len", "lock", "include_lock", "newline",
                            "ignore_special_periods")),
    sd=("split_by_duration", ("max_dur", "even_split", "force_len", "lock", "include_lock", "newline",
                              "ignore_special_periods")),
    mg=("merge_by_gap", ("min_gap", "max_words", "max_chars", "is_sum_max", "lock", "newline")),
    mp=("merge_by_punctuation", ("punctuation", "max_words", "max_chars", "is_sum_max", "lock", "newline")),
    ms=("merge_all_segments", ("record",)),
    cm=("clamp_max", ("medium_factor", "max_dur", "clip_start", "verbose")),
    us=("unlock_all_segments", ()),
    l=("lock", ("startswith", "endswith", "right", "left", "case_sensitive", "strip")),
    rw=("remove_word", ("word", "reassign_ids", "verbose", "record")),
    rs=("remove_segment", ("segment", "reassign_ids", "verbose", "record")),
    p=("pad", ("start_pad", "end_pad", "max_dur", "max_end", "word_level")),
    csl=("convert_to_segment_level", ()),
    isp=("ignore_special_periods", ("enable",)),
)
# editing operations of the reference DSL that are outside this package's scope (DESIGN.md "out of scope")
_UNSUPPORTED = ("rp", "rws", "fg", "ag", "co")


def parse_regroup_algo(result: WhisperResult, regroup_algo: str, include_str: bool = True):
    """result.py:2980-3024: '_' separates operations, '=' introduces arguments, '+' separates positional arguments."""
    if not regroup_algo:
        return []
    calls = regroup_algo.split("_")
    if "da" in calls:
        calls = list(chain.from_iterable(DEFAULT_ALGO.split("_") if c == "da" else [c] for c in calls))
    program = []
    for call in calls:
        key, _, argstr = call.partition("=")
        if key in _UNSUPPORTED:
            raise NotImplementedError(f"regroup operation '{key}' is not provided by stable_ts_amd "
                                      f"(available: {tuple(_OPS)})")
        if key not in _OPS:
            raise NotImplementedError(f"{key} is not one of the available methods: {tuple(_OPS) + _UNSUPPORTED}")
        name, params = _OPS[key]
        values = [_parse_value(a) for a in argstr.split("+")] if argstr else []
        kwargs = {k: v for k, v in zip(params, values) if v is not None}
        shown = None
        if include_str:
            shown = f"{name}(" + ", ".join(f'{k}="{v}"' if isinstance(v, str) else f"{k}={v}"
                                            for k, v in kwargs.items()) + ")"
        program.append((getattr(result, name), kwargs, shown))
    return program


def regroup_default(result: WhisperResult, regroup: Union[bool, str] = True) -> WhisperResult:
    """The hook transcribe()/align() call (original_whisper.py:776-777, alignment.py: ``result.regroup(regroup)``)."""
    return result.regroup(regroup)
