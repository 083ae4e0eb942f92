"""Word-level timestamps from cross-attention + DTW: host glue around ``swx_score`` and ``swx_dtw``.

Mirrors stable_whisper/timing.py (find_alignment_stable :202-306, split_word_tokens :344-392,
add_word_timestamps_stable :411-500).  The teacher-forced decoder pass, the head selection / softmax / z-norm /
median filter / head-mean and the DTW all run on the GPU (csrc/swx_runtime.hip::swx_score, swx_align.hip,
swx_dtw.hip); what stays here is the token<->word bookkeeping, which is string work.

Scope of this round: the reference's default ('legacy') aligner with the model's alignment heads.
``dynamic_heads``, ``aligner='new'`` and ``extra_models`` need the all-heads capture mode.
"""
import string
from dataclasses import dataclass
from itertools import chain
from typing import Callable, List, Optional, Sequence, Union

import numpy as np

from .audio import N_SAMPLES_PER_TOKEN, TOKENS_PER_SECOND

# stable-ts's defaults (stable_whisper/default.py:5-6): upstream whisper's sets plus the CJK corner brackets
PREPEND_PUNCTUATIONS = "\"'“¿([{-「"
APPEND_PUNCTUATIONS = "\"'.。,，!！?？:：”)]}、」"


@dataclass
class WordTiming:
    word: str
    tokens: List[int]
    start: float
    end: float
    probability: float


def merge_punctuations(alignment: List[WordTiming], prepended: str, appended: str):
    """whisper.timing.merge_punctuations (called at timing.py:468): glue punctuation-only words to their neighbour;
    emptied entries keep their slot."""
    j = len(alignment) - 1
    for i in range(len(alignment) - 2, -1, -1):
        prev, nxt = alignment[i], alignment[j]
        if prev.word.startswith(" ") and prev.word.strip() in prepended:
            nxt.word = prev.word + nxt.word
            nxt.tokens = prev.tokens + nxt.tokens
            prev.word, prev.tokens = "", []
        else:
            j = i
    i = 0
    for j in range(1, len(alignment)):
        prev, nxt = alignment[i], alignment[j]
        if not prev.word.endswith(" ") and nxt.word in appended:
            prev.word = prev.word + nxt.word
            prev.tokens = prev.tokens + nxt.tokens
            nxt.word, nxt.tokens = "", []
        else:
            i = j


def _split_tokens(tokens: List[int], tokenizer):
    """timing.py:309-341: group tokens into words by decoding growing prefixes."""
    by_space = getattr(tokenizer, "language_code", tokenizer.language) not in {"zh", "ja", "th", "lo", "my"}
    remaining = tokenizer.decode_with_timestamps(tokens)
    words, groups, pending = [], [], []
    glue = False
    piece = ""
    for t in tokens:
        pending.append(t)
        piece = tokenizer.decode(pending)
        complete = t >= tokenizer.eot
        if not complete:
            complete = remaining[:len(piece)] == piece
            if complete and by_space:
                glue = not (piece.startswith(" ") or piece.strip() in string.punctuation)
        if complete:
            if glue and words:
                words[-1] += piece
                groups[-1].extend(pending)
            else:
                words.append(piece)
                groups.append(pending)
            remaining = remaining[len(piece):]
            pending = []
    if pending:
        words.append(piece if len(remaining) == 0 else remaining)
        groups.append(pending)
    elif remaining:
        words[-1] += remaining
    return words, groups


def split_word_tokens(segments: List[dict], tokenizer, *, padding: Union[str, int, None] = None,
                      split_callback: Callable = None, pad_first_seg: bool = True):
    """timing.py:344-392 without char_split: flat text tokens, (words, word_tokens), segment index per word."""
    if padding is not None:
        padding = tokenizer.encode(padding) if isinstance(padding, str) else [padding]
    flat, seg_of_word, words, groups = [], [], [], []
    for si, seg in enumerate(segments):
        text_only = [t for t in seg["tokens"] if not isinstance(t, int) or t < tokenizer.eot]
        w, g = _split_tokens(text_only, tokenizer) if split_callback is None else split_callback(text_only, tokenizer)
        assert len(w) == len(g), f"word count and token group count do not match, {len(w)} and {len(g)}"
        if (padding is not None and g[0][0] != padding and (len(flat) == 0 or flat[-1] != padding)
                and (pad_first_seg or si != 0)):
            flat.extend(padding)
            words.append(None)
            groups.append(padding)
        seg_of_word.extend([si] * len(w))
        flat.extend(chain.from_iterable(g))
        words.extend(w)
        groups.extend(g)
    return flat, (words, groups), seg_of_word


def pop_empty_alignment(alignment: List[WordTiming], seg_indices: Optional[List[int]] = None):
    """timing.py:395-407: drop the gap-padding pseudo words, remembering the one in front of each segment."""
    if seg_indices is None:
        popped = [alignment.pop(i) for i in reversed(range(len(alignment))) if alignment[i].word is None]
        return list(reversed(popped))
    pos = len(seg_indices)
    removed = {}
    for i in reversed(range(len(alignment))):
        assert pos != -1
        if alignment[i].word is None:
            removed[seg_indices[pos]] = alignment.pop(i)
        else:
            pos -= 1
    return removed


class AlignmentJob:
    """One window's share of a batched alignment call."""

    def __init__(self, tokenizer, text_tokens: List[int], num_samples: int, token_split=None):
        self.tokenizer = tokenizer
        self.text_tokens = list(text_tokens)
        self.num_samples = num_samples
        if token_split is None:
            words, groups = tokenizer.split_to_word_tokens(self.text_tokens + [tokenizer.eot])
        else:
            words, groups = token_split
            words.append(tokenizer.decode([tokenizer.eot]))
            groups.append([tokenizer.eot])
        self.words, self.groups = words, groups
        self.tokens = [*tokenizer.sot_sequence, tokenizer.no_timestamps, *self.text_tokens, tokenizer.eot]
        self.n_frames = round(num_samples / N_SAMPLES_PER_TOKEN)


def find_alignment_batch(model, jobs: Sequence[AlignmentJob], xkv, *, medfilt_width: int = 7, qk_scale: float = 1.0,
                         return_debug: bool = False) -> List[List[WordTiming]]:
    """timing.py:202-306 for W windows at once: scoring pass + alignment matrix + DTW on the device."""
    tok = jobs[0].tokenizer
    eng = model.engine
    probs, neg, T = eng.score(xkv, [j.tokens for j in jobs], [j.n_frames for j in jobs], n_sot=len(tok.sot_sequence),
                              eot=tok.eot, qk_scale=qk_scale, medfilt_width=medfilt_width)
    paths = eng.dtw(neg, [t + 1 for t in T], [j.n_frames for j in jobs])
    out = []
    for w, job in enumerate(jobs):
        text_idx, time_idx = paths[w]
        jumps = np.pad(np.diff(text_idx), (1, 0), constant_values=1).astype(bool)      # timing.py:197
        jump_idx = time_idx[jumps].clip(min=0)                                          # timing.py:198
        jump_times = jump_idx / TOKENS_PER_SECOND
        bounds = np.pad(np.cumsum([len(g) for g in job.groups[:-1]]), (1, 0))           # timing.py:251
        starts, ends = jump_times[bounds[:-1]], jump_times[bounds[1:]]
        p = probs[w]
        wp = [np.mean(p[i:j]) for i, j in zip(bounds[:-1], bounds[1:])]                 # timing.py:292-295
        out.append([WordTiming(a, b, c, d, e) for a, b, c, d, e in zip(job.words, job.groups, starts, ends, wp)])
        if return_debug:
            job.debug = dict(path=(text_idx, time_idx), jump_idx=jump_idx, token_probs=p)
    return out


def add_word_timestamps_batch(*, model, tokenizer, windows: Sequence[dict], xkv,
                              prepend_punctuations: str = None, append_punctuations: str = None,
                              min_word_dur: float = 0.1, split_callback: Callable = None,
                              gap_padding: Optional[str] = " ...", pad_first_seg: bool = True,
                              medfilt_width: int = 7, qk_scale: float = 1.0):
    """timing.py:411-500 for several windows at once.  windows[w] = dict(segments=[...], num_samples=int); the window
    order matches the batch order inside `xkv`.  Mutates segments[i]['words'] / ['start'] / ['end'] in place."""
    prepend_punctuations = PREPEND_PUNCTUATIONS if prepend_punctuations is None else prepend_punctuations
    append_punctuations = APPEND_PUNCTUATIONS if append_punctuations is None else append_punctuations
    min_word_dur = min_word_dur or 0
    assert all(len(wd["segments"]) > 0 for wd in windows)
    jobs, seg_maps = [], []
    for wd in windows:
        for seg in wd["segments"]:
            seg["words"] = []
        flat, token_split, seg_of_word = split_word_tokens(wd["segments"], tokenizer, padding=gap_padding,
                                                           split_callback=split_callback, pad_first_seg=pad_first_seg)
        jobs.append(AlignmentJob(tokenizer, flat, wd["num_samples"], token_split))
        seg_maps.append(seg_of_word)
    alignments = find_alignment_batch(model, jobs, xkv, medfilt_width=medfilt_width, qk_scale=qk_scale)
    for wd, alignment, seg_of_word in zip(windows, alignments, seg_maps):
        segments = wd["segments"]
        lead = pop_empty_alignment(alignment, seg_of_word)
        merge_punctuations(alignment, prepend_punctuations, append_punctuations)
        offset = segments[0]["seek"]
        assert len(alignment) == len(seg_of_word)
        for si, timing in zip(seg_of_word, alignment):
            if len(timing.tokens) == 0:
                continue
            start, end = timing.start, timing.end
            if len(segments[si]["words"]) == 0 and (end - start) < min_word_dur and si in lead:
                start = lead[si].start          # timing.py:477-483: borrow the gap-padding start
            segments[si]["words"].append(dict(word=timing.word, start=round(offset + start, 3),
                                              end=round(offset + end, 3), probability=timing.probability,
                                              tokens=timing.tokens))
        for seg in segments:
            if seg["words"]:
                seg["start"] = seg["words"][0]["start"]
                seg["end"] = seg["words"][-1]["end"]
