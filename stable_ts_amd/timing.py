"""Word-level timestamps from cross-attention + DTW: host glue around ``swx_score`` and ``swx_dtw``.

Mirrors stable_whisper/timing.py (find_alignment_stable :202-306, split_word_tokens :344-392,
add_word_timestamps_stable :411-500).  The teacher-forced decoder pass, the head selection / softmax / z-norm /
median filter / head-mean and the DTW all run on the GPU (csrc/swx_runtime.hip::swx_score, swx_align.hip,
swx_dtw.hip); what stays here is the token<->word bookkeeping, which is string work.

The default ('legacy') aligner with the model's alignment heads is one fused device call (``swx_score``).  The
head-selection variants -- ``dynamic_heads`` (timing.py:87-103), ``aligner='new'`` (timing.py:115-163) and
``extra_models`` (timing.py:177-189) -- are kernels too since round 3 (csrc/swx_headsel.hip): the pass keeps the
cross-attention queries of every layer (``swx_score_q``, 8-37 MB) instead of every head's scores (0.4-1.7 GB in the
reference), a head's score row is recomputed from q and the resident cross-K where it is needed; z-normalisation, median
filter, head mean and DTW are the kernels of the default path.
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
    # most words are one or two tokens: the decoded form of a single pending token is looked up (same string as decode([t]),
    # memoised per tokenizer) instead of going through the list filter + join of `decode` every time
    # (a slotted or wrapped tokenizer has no instance dict: no memo, plain decode)
    d_ = getattr(tokenizer, "__dict__", None)
    single = d_.setdefault("_single_piece", {}) if isinstance(d_, dict) else {}
    words, groups, pending = [], [], []
    glue = False
    piece = ""
    for t in tokens:
        pending.append(t)
        if len(pending) == 1 and type(t) is int:
            piece = single.get(t)
            if piece is None:
                piece = single[t] = tokenizer.decode(pending)
        else:
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


def parse_dynamic_heads(dynamic_heads) -> tuple:
    """timing.py:255-268: True -> 6 heads, int -> that many, "k,n" -> k heads refined over n iterations"""
    if not dynamic_heads:
        return None, None
    if dynamic_heads is True:
        return 6, None
    if isinstance(dynamic_heads, int):
        return dynamic_heads, None
    assert "," in dynamic_heads
    k, n = dynamic_heads.split(",")[:2]
    return int(k), int(n)


def _find_alignment_variants(model, jobs, xkv, *, medfilt_width, qk_scale, dynamic_heads, aligner, extra_models, mel,
                             return_debug):
    """timing.py:166-198 + 202-306 for the head-selection variants, one window at a time.  The per-head arithmetic runs in
    csrc/swx_headsel.hip through the engine: ``score_q`` (the pass, keeping the cross-attention queries), ``heads_dynamic``
    (timing.py:87-112), ``heads_new`` (timing.py:115-163), ``pool_matrices`` (timing.py:177-189); what stays here is the
    control flow of the reference (which models, how many refinement iterations, which probabilities are averaged)."""
    from .transcribe import _xkv_select
    assert isinstance(aligner, dict) or aligner in ("new", "legacy"), f'aligner must be "new"/"legacy", got "{aligner}"'
    if extra_models and (bad := set(map(type, extra_models)) - {type(model)}):
        raise NotImplementedError(f"Got unsupported model type(s): {bad}")
    tok = jobs[0].tokenizer
    n_sot, eot = len(tok.sot_sequence), tok.eot
    count, iterations = parse_dynamic_heads(dynamic_heads)
    new = aligner != "legacy"
    if not new and getattr(model, "missing_alignment_heads", False) and not count:
        count = 6
    extras = []
    if extra_models and not new:
        if mel is None:
            raise ValueError("extra_models need the windows' log-mel (each model encodes the audio itself)")
        extras = [(m, m.cross_kv(m.encoder(mel))) for m in extra_models]

    def legacy_matrix(m, xkv_w, job, state, jump):
        """one model's NEGATED head mean for the text-token rows + its token probabilities + how many heads went into it"""
        eng = m.engine
        if count:
            st = state.get(id(m)) or state.setdefault(id(m), eng.score_q(xkv_w, job.tokens, n_sot=n_sot, eot=eot))
            neg = eng.heads_dynamic(st, job.n_frames, count=count, qk_scale=qk_scale, medfilt_width=medfilt_width, jump_indices=jump)
            return neg, st["probs"], count
        p, neg, T = eng.score(xkv_w, [job.tokens], [job.n_frames], n_sot=n_sot, eot=eot, qk_scale=qk_scale, medfilt_width=medfilt_width)
        return neg[0, :T[0] + 1].contiguous(), p[0], eng.n_alignment_heads

    out = []
    for w, job in enumerate(jobs):
        eng = model.engine
        xkv_w = _xkv_select(model, xkv, [w])
        jump, probs, state = None, None, {}
        for _ in range(iterations or 1):
            if new:
                st = state.get("new") or state.setdefault("new", eng.score_q(xkv_w, job.tokens, n_sot=n_sot, eot=eot))
                neg = eng.heads_new(st, job.n_frames, qk_scale=qk_scale, medfilt_width=medfilt_width,
                                    **({k: v for k, v in aligner.items()} if isinstance(aligner, dict) else {}))
                probs = st["probs"]
            else:
                neg, p, n_heads = legacy_matrix(model, xkv_w, job, state, jump)
                probs = p if probs is None else probs          # the main model's pass is cached across iterations
                if extras:
                    negs, heads, extra_probs = [neg], [n_heads], []
                    for m, xkv_m in extras:
                        ne, pe, he = legacy_matrix(m, _xkv_select(m, xkv_m, [w]), job, state, None)
                        negs.append(ne)
                        heads.append(he)
                        extra_probs.append(pe)
                    neg = eng.pool_matrices(negs, heads)
                    import torch
                    probs = torch.tensor(extra_probs + [probs]).mean(dim=0).tolist()       # timing.py:183-189 (host, a few floats)
            neg = neg[None, :, :job.n_frames].contiguous() if neg.shape[-1] != job.n_frames else neg[None].contiguous()
            (text_idx, time_idx), = eng.dtw(neg, [neg.shape[1]], [neg.shape[2]])
            jumps = np.pad(np.diff(text_idx), (1, 0), constant_values=1).astype(bool)
            jump = time_idx[jumps].clip(min=0)
        out.append((probs, text_idx, time_idx))
    return out


def _word_means(pa: np.ndarray, bounds: np.ndarray) -> list:
    """``[np.mean(pa[i:j]) for i, j in zip(bounds[:-1], bounds[1:])]`` (timing.py:292-295) without ~2 000 ``np.mean`` calls per
    10-minute pass (5 ms of host time with the device idle).  Below 8 elements numpy's sum adds left to right starting from 0.0
    (its pairwise blocking starts at 8), so the words of k = 1..7 tokens are summed column by column, all words of one k at a
    time -- the same additions in the same order, bit for bit (randomised check in tests/test_host_cpu.py); longer or empty
    words (NaN + numpy's warning, as upstream) take ``np.mean`` itself."""
    cnt = np.diff(bounds)
    out = [None] * len(cnt)
    for k in range(1, 8):
        sel = np.flatnonzero(cnt == k)
        if sel.size:
            base = bounds[sel]
            acc = pa[base]                                   # 0.0 + a0 == a0
            for t in range(1, k):
                acc = acc + pa[base + t]
            for i, v in zip(sel.tolist(), acc / k):
                out[i] = v
    for i in np.flatnonzero((cnt < 1) | (cnt >= 8)).tolist():
        out[i] = np.mean(pa[bounds[i]:bounds[i + 1]])
    return out


def find_alignment_batch(model, jobs: Sequence[AlignmentJob], xkv, *, medfilt_width: int = 7, qk_scale: float = 1.0,
                         dynamic_heads=None, aligner: Union[str, dict] = "legacy", extra_models: Optional[list] = None,
                         mel=None, return_debug: bool = False, started=None) -> List[List[WordTiming]]:
    """timing.py:202-306 for W windows at once: scoring pass + alignment matrix + DTW on the device."""
    tok = jobs[0].tokenizer
    eng = model.engine
    import time
    _t0 = time.perf_counter()
    if dynamic_heads or extra_models or aligner != "legacy" or getattr(model, "missing_alignment_heads", False):
        res = _find_alignment_variants(model, jobs, xkv, medfilt_width=medfilt_width, qk_scale=qk_scale,
                                       dynamic_heads=dynamic_heads, aligner=aligner, extra_models=extra_models, mel=mel,
                                       return_debug=return_debug)
        probs, paths = [r[0] for r in res], [(r[1], r[2]) for r in res]
    else:
        if started is not None:              # the scoring pass was enqueued before the words were split (same tokens: checked)
            assert started["tokens"] == [j.tokens for j in jobs]
            probs, neg, T = eng.score_finish(started["handle"])
        else:
            probs, neg, T = eng.score(xkv, [j.tokens for j in jobs], [j.n_frames for j in jobs], n_sot=len(tok.sot_sequence),
                                      eot=tok.eot, qk_scale=qk_scale, medfilt_width=medfilt_width)
        paths = eng.dtw(neg, [t + 1 for t in T], [j.n_frames for j in jobs])
    from . import transcribe as _tr
    if _tr.PHASE_TIMES is not None:                       # diagnostic: device part of the word-timestamp stage
        import time
        _tr.PHASE_TIMES["  of which device (score + a7 + DTW, synchronous copy-out)"] = (
            _tr.PHASE_TIMES.get("  of which device (score + a7 + DTW, synchronous copy-out)", 0.0) + time.perf_counter() - _t0)
    out = []
    for w, job in enumerate(jobs):
        text_idx, time_idx = paths[w]
        jumps = np.pad(np.diff(text_idx), (1, 0), constant_values=1).astype(bool)      # timing.py:197
        jump_idx = time_idx[jumps].clip(min=0)                                          # timing.py:198
        jump_times = jump_idx / TOKENS_PER_SECOND
        bounds = np.pad(np.cumsum([len(g) for g in job.groups[:-1]]), (1, 0))           # timing.py:251
        starts, ends = jump_times[bounds[:-1]], jump_times[bounds[1:]]
        p = probs[w]
        pa = np.asarray(p, dtype=np.float64)         # (np.mean of a list slice converts the slice every time: same values)
        wp = _word_means(pa, bounds)                                                    # timing.py:292-295
        out.append([WordTiming(a, b, c, d, e) for a, b, c, d, e in zip(job.words, job.groups, starts, ends, wp)])
        if return_debug:
            job.debug = dict(path=(text_idx, time_idx), jump_idx=jump_idx, token_probs=p)
    return out


def add_word_timestamps_batch(*, model, tokenizer, windows: Sequence[dict], xkv,
                              prepend_punctuations: str = None, append_punctuations: str = None,
                              min_word_dur: float = 0.1, split_callback: Callable = None,
                              gap_padding: Optional[str] = " ...", pad_first_seg: bool = True,
                              medfilt_width: int = 7, qk_scale: float = 1.0, dynamic_heads=None,
                              aligner: Union[str, dict] = "legacy", extra_models: Optional[list] = None, mel=None):
    """timing.py:411-500 for several windows at once.  windows[w] = dict(segments=[...], num_samples=int); the window
    order matches the batch order inside `xkv`.  Mutates segments[i]['words'] / ['start'] / ['end'] in place."""
    prepend_punctuations = PREPEND_PUNCTUATIONS if prepend_punctuations is None else prepend_punctuations
    append_punctuations = APPEND_PUNCTUATIONS if append_punctuations is None else append_punctuations
    min_word_dur = min_word_dur or 0
    assert all(len(wd["segments"]) > 0 for wd in windows)
    # The scoring pass only needs the token sequences, which are known before the words are: on the default path enqueue it
    # first and split the words (tokenizer decoding, ~0.6 ms per window of host time) while the device runs it.
    started = None
    eng = getattr(model, "engine", None)
    if (split_callback is None and not dynamic_heads and not extra_models and aligner == "legacy"
            and not getattr(model, "missing_alignment_heads", False) and hasattr(eng, "score_start")):
        pad = None if gap_padding is None else (tokenizer.encode(gap_padding) if isinstance(gap_padding, str) else [gap_padding])
        toks, frames = [], []
        for wd in windows:
            flat = []
            for si, seg in enumerate(wd["segments"]):
                text_only = [t for t in seg["tokens"] if not isinstance(t, int) or t < tokenizer.eot]
                if (pad is not None and text_only and text_only[0] != pad and (len(flat) == 0 or flat[-1] != pad)
                        and (pad_first_seg or si != 0)):
                    flat.extend(pad)
                flat.extend(text_only)
            toks.append([*tokenizer.sot_sequence, tokenizer.no_timestamps, *flat, tokenizer.eot])
            frames.append(round(wd["num_samples"] / N_SAMPLES_PER_TOKEN))
        started = dict(tokens=toks, handle=eng.score_start(xkv, toks, frames, n_sot=len(tokenizer.sot_sequence),
                                                           eot=tokenizer.eot, qk_scale=qk_scale, medfilt_width=medfilt_width))
    jobs, seg_maps = [], []
    for wd in windows:
        for seg in wd["segments"]:
            seg["words"] = []
        flat, token_split, seg_of_word = split_word_tokens(wd["segments"], tokenizer, padding=gap_padding,
                                                           split_callback=split_callback, pad_first_seg=pad_first_seg)
        jobs.append(AlignmentJob(tokenizer, flat, wd["num_samples"], token_split))
        seg_maps.append(seg_of_word)
    if started is not None and started["tokens"] != [j.tokens for j in jobs]:
        started = None                         # cannot happen with the built-in splitter; a mismatch just costs a second pass
    alignments = find_alignment_batch(model, jobs, xkv, medfilt_width=medfilt_width, qk_scale=qk_scale,
                                      dynamic_heads=dynamic_heads, aligner=aligner, extra_models=extra_models, mel=mel,
                                      started=started)
    for wd, alignment, seg_of_word in zip(windows, alignments, seg_maps):
        segments = wd["segments"]
        lead = pop_empty_alignment(alignment, seg_of_word)
        merge_punctuations(alignment, prepend_punctuations, append_punctuations)
        offset = segments[0]["seek"]
        assert len(alignment) == len(seg_of_word)
        kept, first_of_seg, t_se = [], set(), []
        for si, timing in zip(seg_of_word, alignment):
            if len(timing.tokens) == 0:
                continue
            start, end = timing.start, timing.end
            if si not in first_of_seg and (end - start) < min_word_dur and si in lead:
                start = lead[si].start          # timing.py:477-483: borrow the gap-padding start
            first_of_seg.add(si)
            kept.append((si, timing))
            t_se.append((start, end))
        if kept:
            # round(offset + t, 3) of numpy scalars, all words of the window in one call: the same additions and the same
            # numpy rounding (np.float64.__round__ is np.round), without 2 x 1.6 us of scalar dispatch per word
            if all(isinstance(v, np.floating) for se in t_se for v in se):
                r = np.round(offset + np.asarray(t_se, dtype=np.float64), 3)
                rounded = [(r[i, 0], r[i, 1]) for i in range(len(kept))]
            else:
                rounded = [(round(offset + a, 3), round(offset + b, 3)) for a, b in t_se]
            for (si, timing), (rs, re_) in zip(kept, rounded):
                segments[si]["words"].append(dict(word=timing.word, start=rs, end=re_, probability=timing.probability,
                                                  tokens=timing.tokens))
        for seg in segments:
            if seg["words"]:
                seg["start"] = seg["words"][0]["start"]
                seg["end"] = seg["words"][-1]["end"]
