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


# -------------------------------------------------------------------------------------------------- word edits
def _norm_words(texts: List[str], strip: bool, ignore_punctuations: str, case_sensitive: bool) -> List[str]:
    if strip:
        texts = [w.strip() for w in texts]
    if ignore_punctuations:
        ptn = f"[{ignore_punctuations}]+$"
        texts = [re.sub(ptn, "", w) for w in texts]
    if not case_sensitive:
        texts = [w.lower() for w in texts]
    return texts


def remove_repetition(result: WhisperResult, max_words: int, case_sensitive: bool, strip: bool, ignore_punctuations: str,
                      extend_duration: bool):
    """result.py:2238-2326 -- delete immediate repeats of runs of 1..max_words words (right to left), optionally
    stretching the word before the repeat over it; of two copies the one with the longer text survives in place."""
    for count in range(1, max_words + 1):
        words = result.all_words()
        if len(words) < 2:
            return
        texts = _norm_words([w.word for w in words], strip, ignore_punctuations, case_sensitive)
        resume_at = None
        for i in reversed(range(count * 2, len(texts) + 1)):
            if resume_at is not None:
                if resume_at != i:
                    continue
                resume_at = None
            s0 = i - count
            if texts[s0 - count:s0] != texts[s0:i]:
                continue
            resume_at = s0
            if extend_duration:
                words[s0 - 1].end = words[i - 1].end
            for j in reversed(range(s0, i)):
                result.remove_word(words[j], False, verbose=False, record=False)
            for i0, i1 in zip(range(s0 - count, s0), range(s0, i)):
                if len(words[i0].word) < len(words[i1].word):
                    words[i1].start, words[i1].end = words[i0].start, words[i0].end
                    result.segments[words[i0].segment_id].words[words[i0].id] = words[i1]
        result.remove_no_word_segments(reassign_ids=False)
    result.reassign_ids()


def remove_words_by_str(result: WhisperResult, words, case_sensitive: bool, strip: bool, ignore_punctuations: str,
                        min_prob: Optional[float], filters: Optional[Callable]):
    """result.py:2328-2405."""
    all_words = result.all_words()
    texts = _norm_words([w.word for w in all_words], strip, ignore_punctuations, case_sensitive)
    targets = None if words is None else _norm_words(list(words), strip, ignore_punctuations, case_sensitive)
    for i, t in reversed(list(enumerate(texts))):
        if not (targets is None or any(t == x for x in targets)):
            continue
        w = all_words[i]
        if (min_prob is None or w.probability is None or min_prob > w.probability) and (filters is None or filters(w)):
            result.remove_word(w, False, verbose=False, record=False)
    result.remove_no_word_segments()


def fill_in_gaps(result: WhisperResult, other: WhisperResult, min_gap: float, case_sensitive: bool, strip: bool,
                 ignore_punctuations: str):
    """result.py:2407-2513 -- words of `other` that fall into gaps (> min_gap) between this result's segments are
    inserted as new segments; a gap word equal to the neighbouring word only extends that neighbour."""
    def key(w: str) -> str:
        return _norm_words([w], strip, ignore_punctuations, case_sensitive)[0]

    segs = result.segments
    pairs = [(-1, (None, segs[0]))] + list(enumerate(zip(segs[:-1], segs[1:])))
    pairs.append((pairs[-1][0] + 1, (segs[-1], None)))
    for i, (left, right) in reversed(pairs):
        first = None if left is None else left.words[-1]
        last = None if right is None else right.words[0]
        start = other[0].start if first is None else first.end
        end = other[-1].end if last is None else last.start
        if end - start <= min_gap:
            continue
        gap = other.get_content_by_time((start, end))
        if first is not None and gap and key(first.word) == key(gap[0].word):
            first.end = gap[0].end
            gap = gap[1:]
        if last is not None and gap and key(last.word) == key(gap[-1].word):
            last.start = gap[-1].start
            gap = gap[:-1]
        if not gap:
            continue
        if last is not None and last.start < gap[-1].end:
            last.start = gap[-1].end
        new_segs = [other[gap[0].segment_id].spawn([])]
        for j, w in enumerate(gap):
            c = w.copy(copy_tokens=True)
            if j == 0 and first is not None and first.end > gap[0].start:
                c.start = first.end
            if new_segs[-1].id != w.segment_id:
                new_segs.append(other[w.segment_id].spawn([]))
            new_segs[-1].words.append(c)
        result.segments = result.segments[:i + 1] + new_segs + result.segments[i + 1:]
    result.reassign_ids()


def adjust_gaps(result: WhisperResult, duration_threshold: float, one_section: bool):
    """result.py:2515-2628 -- move the end of each segment / start of the next to the boundaries of the dominant
    non-speech section(s) detected between them."""
    if duration_threshold > 1:
        raise ValueError(f"``duration_threshold`` must be at most 1.0 but got {duration_threshold}")
    segs = result.segments
    ns_idx = 0
    for si in range(-1, len(segs)):
        curr = None if si == -1 else segs[si]
        nxt = None if curr is segs[-1] else segs[si + 1]
        cs = ce = ns = ne = None
        if result.has_words:
            if curr is None:
                d = np.median([w.duration for w in nxt.words]) * 2
                cs = ce = max(nxt.start - d, 0)
            if nxt is None:
                d = np.median([w.duration for w in curr.words]) * 2
                ns = ne = curr.end + d
            if curr is not None:
                curr = curr.words[-1]
            if nxt is not None:
                nxt = nxt.words[0]
        else:
            if curr is None:
                cs = ce = max(nxt.start - nxt.duration, 0)
            if nxt is None:
                ns = ne = curr.end + curr.duration
        cs = curr.start if cs is None else cs
        ce = curr.end if ce is None else ce
        ns = nxt.start if ns is None else ns
        ne = nxt.end if ne is None else ne
        sections: List[Tuple[float, float]] = []
        for ns_idx in range(ns_idx, len(result.nonspeech_sections)):
            sec = result.nonspeech_sections[ns_idx]
            a, b = sec["start"], sec["end"]
            if cs < (b if curr is None else a) and (a if nxt is None else b) < ne:
                sections.append((a, b))
            if ns < a:
                break
        if not sections:
            continue
        durs = np.array([b - a for a, b in sections])
        order = np.argsort(durs)
        durs = durs[order]
        ok = durs / durs[-1] >= duration_threshold
        if not np.any(ok):
            continue
        order = order[ok]
        c_scores = np.array([abs(sections[k][0] - ce) for k in order])
        n_scores = np.array([abs(sections[k][1] - ns) for k in order])
        if one_section:
            bc = bn = order[np.argmin(c_scores + n_scores)]
        else:
            bc, bn = order[np.argmin(c_scores)], order[np.argmin(n_scores)]
            if bc > bn:
                bc = bn = order[np.argmin(c_scores + n_scores)]
        new_end = sections[bc][0]
        if curr is not None and cs < new_end:
            curr.end = new_end
        new_start = sections[bn][1]
        if nxt is not None and new_start < ne:
            nxt.start = new_start


_OPERATORS = {"==": lambda a, b: a == b, ">": lambda a, b: a > b, ">=": lambda a, b: a >= b, "<": lambda a, b: a < b,
              "<=": lambda a, b: a <= b, "is": lambda a, b: a is b, "in": lambda a, b: a in b, "start": str.startswith,
              "end": str.endswith}
_ACTIONS = ("mergeleft", "mergeright", "merge", "lockright", "lockleft", "lock", "splitright", "splitleft", "split",
            "remove")


def custom_operation(result: WhisperResult, key: str, operator, value, method, word_level: Optional[bool]):
    """result.py:2653-2891 -- apply merge / lock / split / remove (or a callable) to every word or segment whose
    attribute `key` satisfies `operator(attribute, value)`, right to left.  Returns the strings for the history entry."""
    if result.has_words:
        if word_level is None:
            word_level = True
    elif word_level:
        raise ValueError("result is missing word timestamps and not compatible with ``word_level=True``")
    value = result._get_content(value, strict=False)
    method = result._get_content(method)
    builtin = isinstance(method, str)
    if builtin:
        if method not in _ACTIONS:
            raise ValueError(f"invalid method: '{method}'. Valid methods: {method}")
    elif not callable(method):
        raise TypeError(f"'{type(method)}' object is not callable")
    key = key.replace(" ", "_")
    operator = result._get_content(operator)
    if isinstance(operator, str):
        if operator not in _OPERATORS:
            raise ValueError(f"invalid operator: '{operator}'. Valid operators: {tuple(_OPERATORS)}")
        operator_str, operator = operator, _OPERATORS[operator]
    else:
        operator_str = result._store_content(operator)
    method_str = method if builtin else result._store_content(method)
    if builtin and method.startswith("split"):
        if word_level is None:
            raise ValueError("Segment-level result is not compatible with split actions.")
        if not word_level:
            raise ValueError("``word_level=False`` is not compatible with split actions.")

    def act(si: int, wi: Optional[int]):
        if not builtin:
            return method(result, si, wi)
        seg = result.segments[si]
        if method.startswith("merge"):
            pairs = []
            if method in ("mergeright", "merge") and not (si + 1 >= len(result.segments) or
                                                          (wi is not None and wi != len(seg.words) - 1)):
                pairs.append((si, si + 1))
            if method in ("mergeleft", "merge") and not (si == 0 or (wi is not None and wi != 0)):
                pairs.append((si - 1, si))
            for a, b in pairs:
                result.add_segments(a, b, inplace=True, reassign_ids=False)
        elif method.startswith("lock"):
            target = seg if wi is None else seg.words[wi]
            if method in ("lockright", "lock"):
                target.lock_right()
            if method in ("lockleft", "lock"):
                target.lock_left()
        elif method == "splitright":
            if wi != len(seg.words) + 1:
                result.split_segment_by_index(seg, wi, reassign_ids=False)
        elif method == "splitleft":
            if wi != 0:
                result.split_segment_by_index(seg, wi - 1, reassign_ids=False)
        elif method == "split":
            idx = ([wi - 1] if wi != 0 else []) + ([wi] if wi < len(seg.words) + 1 else [])
            result.split_segment_by_index(seg, idx, reassign_ids=False)
        elif wi is None:
            result.remove_segment(seg, reassign_ids=False, record=False, verbose=False)
        else:
            result.remove_word(seg.words[wi], reassign_ids=False, record=False, verbose=False)

    if key.startswith("len="):
        get = lambda o: len(getattr(o, key[4:]))          # noqa: E731
    elif key == "":
        get = lambda o: o                                  # noqa: E731
    else:
        get = lambda o: getattr(o, key)                    # noqa: E731
    if isinstance(value, str) and (value.startswith("all=") or value.startswith("any=")):
        check = any if value.startswith("any=") else all
        values = [v.replace("\\,", ",") for v in re.split(r"(?<!\\),", value[4:])]
        ok = lambda o: check(operator(get(o), v) for v in values)      # noqa: E731
    else:
        ok = lambda o: operator(get(o), value)             # noqa: E731
    for si in range(len(result.segments) - 1, -1, -1):
        if word_level:
            for wi in range(len(result.segments[si].words) - 1, -1, -1):
                if ok(result.segments[si].words[wi]):
                    act(si, wi)
        elif ok(result.segments[si]):
            act(si, None)
    result.reassign_ids()
    if isinstance(value, bool):
        value = f"<{value}>"
    elif not isinstance(value, (str, int, float)):
        value = result._store_content(value)
    return key.replace("_", " "), operator_str, value, method_str, int(word_level)


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
    rp=("remove_repetition", ("max_words", "case_sensitive", "strip", "ignore_punctuations", "extend_duration", "verbose")),
    rws=("remove_words_by_str", ("words", "case_sensitive", "strip", "ignore_punctuations", "min_prob", "filters",
                                 "verbose")),
    fg=("fill_in_gaps", ("other_result", "min_gap", "case_sensitive", "strip", "ignore_punctuations", "verbose")),
    ag=("adjust_gaps", ("duration_threshold", "one_section")),
    co=("custom_operation", ("key", "operator", "value", "method", "word_level")),
)
_UNSUPPORTED = ()


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
