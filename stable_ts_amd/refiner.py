"""``refine()``: tighten existing word timestamps by muting audio and watching the token probabilities (SURVEY.md 8f
row 3).

Behavioural contract = ``stable_whisper/non_whisper/refinement.py::Refiner`` (:13-487).  For every word the start is
moved as late (and the end as early) as possible while the probability of the word's first (last) token, computed by a
teacher-forced pass over the partly muted audio, stays acceptable.  Per group of words (<= ``max_inference_tokens``
tokens, <= 30 s) this is a bisection on the mute boundary of every word at once; two copies of the audio (even / odd
words) are evaluated per inference call so that neighbouring words do not mute each other (:359-475).

Generic over ``inference_func(audio[2, n], tokens) -> probabilities [2, len(tokens)] or [2, len(tokens), vocab]`` (seam
B3): the GPU path plugs in ``stable_ts_amd.alignment.make_refinement_func`` (mel -> encoder -> one teacher-forced decoder
pass for both copies -> softmax over the text vocabulary on the device), the CPU test plugs the same synthetic function
into this class and into the reference's ``Refiner`` and requires identical timestamps.

The bisection state lives in int32 sample arrays exactly like the reference's (`lo`, `hi`, `mid` per word), including
its quirks, which are part of the observable behaviour: the "original" probability a word is compared with is
overwritten by the latest probe (``new_probs`` aliases ``orig_probs``, :405, 472), and the audio copy a word mutes is
looked up in the per-TOKEN row table with the WORD index (:424).
"""
import copy
from typing import Callable, Iterator, List, Optional, Tuple

import numpy as np
import torch

from .result import WhisperResult, WordTiming


class Refiner:
    def __init__(self, inference_func: Callable, sample_rate: int = 16000, max_segment_length="30s",
                 max_inference_tokens: int = 100, *, steps: str = "se", rel_prob_decrease: float = .03,
                 abs_prob_decrease: float = .05, rel_rel_prob_decrease: Optional[float] = None,
                 prob_threshold: float = .5, rel_dur_change: Optional[float] = .5,
                 abs_dur_change: Optional[float] = None, word_level: bool = True, precision: Optional[float] = None,
                 progress_callback: Optional[Callable] = None, **unsupported):
        steps = steps or "se"
        bad = steps.replace("s", "").replace("e", "")
        if bad:
            raise ValueError(f'Invalid step(s): {", ".join(bad)}')
        if isinstance(max_segment_length, str):
            if not max_segment_length.endswith("s"):
                raise ValueError(f'expect string ``max_segment_length`` to end with "s" but got "{max_segment_length}"')
            self.max_segment_seconds = float(max_segment_length[:-1])
        else:
            self.max_segment_seconds = max_segment_length / sample_rate
        for k in ("denoiser", "only_voice_freq"):
            if unsupported.pop(k, None):
                raise NotImplementedError(f"{k} is outside this package's scope (DESIGN.md section 7)")
        for k in ("verbose", "denoiser_options", "all_options", "only_ffmpeg"):
            unsupported.pop(k, None)
        if unsupported:
            raise TypeError(f"unexpected keyword argument(s): {', '.join(unsupported)}")
        self.inference_func = inference_func
        self.sample_rate = sample_rate
        self.max_inference_tokens = max_inference_tokens
        self.steps = steps
        self.precision = 0.1 if precision is None else precision
        self.sample_precision = max(round(self.precision * self.sample_rate), 2)
        self.rel_prob_decrease, self.abs_prob_decrease = rel_prob_decrease, abs_prob_decrease
        self.rel_rel_prob_decrease, self.prob_threshold = rel_rel_prob_decrease, prob_threshold
        self.rel_dur_change, self.abs_dur_change = rel_dur_change, abs_dur_change
        self.word_level = word_level
        self.progress_callback = progress_callback
        self._audio = torch.tensor([])

    # ----------------------------------------------------------------------------------------------- driver
    def refine(self, audio: torch.Tensor, result: WhisperResult, inplace: bool = True,
               encode: Optional[Callable] = None) -> WhisperResult:
        if result:
            if not result.has_words:
                raise RuntimeError("cannot refine result with missing word-timestamps")
            if not all(w.tokens for w in result.all_words()):
                if encode is None:
                    raise RuntimeError("result must have tokens or provide tokenization function to ``encode``")
                for w in result.all_words():
                    w.tokens = encode(w.word)
        if not inplace:
            result = copy.deepcopy(result)
        self._audio = torch.as_tensor(audio, dtype=torch.float32).detach().cpu()
        for n, step in enumerate(self.steps, 1):
            self._refine(result, step)
            if self.progress_callback is not None:
                total = round(self._audio.size(-1) / self.sample_rate, 2)
                self.progress_callback(round(total * n / len(self.steps), 2), total)
        result.reassign_ids()
        return result

    # ------------------------------------------------------------------------------------------- grouping
    def _groups(self, result: WhisperResult, total_duration: float) -> Iterator[Tuple[List[WordTiming], list, list, np.ndarray]]:
        """Consecutive words sharing one inference call, with the earliest start / latest end each word may move to
        (:220-271): bounded by ``rel_dur_change`` x its duration, ``abs_dur_change``, the neighbouring words, and 14.5 s."""
        words = result.all_words()
        edge = np.array([1 if i == 0 else (2 if i == len(s.words) - 1 else 0) for s in result.segments
                         for i in range(len(s.words))])
        lo = [max(0 if self.abs_dur_change is None else (w.start - self.abs_dur_change),
                  0 if self.rel_dur_change is None else (w.start - w.duration * self.rel_dur_change),
                  0 if i == 0 else max(words[i - 1].end, w.end - 14.5, 0))
              for i, w in enumerate(words)]
        hi = [min(total_duration if self.abs_dur_change is None else (w.end + self.abs_dur_change),
                  total_duration if self.rel_dur_change is None else (w.end + w.duration * self.rel_dur_change),
                  total_duration if i == len(words) else min(words[i].start, w.start + 14.5, total_duration))
              for i, w in enumerate(words, 1)]
        t0 = lo[0]
        first = 0
        g_words, g_lo, g_hi, n_tok = [], [], [], 0
        for i, w in enumerate(words, 1):
            if (hi[0] - t0 > self.max_segment_seconds) or (n_tok + len(w.tokens) > self.max_inference_tokens):
                if g_words:
                    yield g_words, g_lo, g_hi, edge[first:first + len(g_words)]
                    g_words, g_lo, g_hi = [], [], []
                t0 = lo[0]
                first = i - 1
                n_tok = 0
            g_words.append(w)
            g_lo.append(lo.pop(0))
            g_hi.append(hi.pop(0))
            n_tok += len(w.tokens)
            if i == len(words):
                yield g_words, g_lo, g_hi, edge[first:first + len(g_words)]

    def _samples(self, seconds, offset: float) -> np.ndarray:
        return ((np.asarray(seconds) - offset) * self.sample_rate).round().astype(np.int32)

    # ----------------------------------------------------------------------------------------------- probing
    def _probe(self, audio2: torch.Tensor, text_tokens: List[int], word_tokens: List[List[int]], rows: List[int],
               at_end: bool):
        """One inference call -> per word the probability of its first (last, for end refinement) token in the audio
        copy that word owns, and that token's rank among the vocabulary when the function returns a distribution."""
        p: torch.Tensor = self.inference_func(audio2, text_tokens)
        if p.size(0) != 2:
            raise RuntimeError(f"expected dim 0 to be length of 2 but got {p.size(0)}")
        if p.size(1) != len(text_tokens):
            raise RuntimeError(f"expected dim 1 to be length of {len(text_tokens)} but got {p.size(1)}")
        if p.ndim not in (2, 3):
            raise RuntimeError(f"expected inference_func output to have 2 or 3 dimensions but got {p.ndim}")
        pos = torch.arange(len(text_tokens))
        bounds = np.pad(np.cumsum([len(t) for t in word_tokens]), (1, 0))
        pick = [(j - 1 if at_end else i) for i, j in zip(bounds[:-1], bounds[1:])]
        if p.ndim == 2:
            tok_p = p[rows, pos].tolist()
            ranks = [0] * len(word_tokens)
        else:
            tok_p = p[:, pos, text_tokens][rows, pos].tolist()
            dist = p[:, pos][rows, pos]                                 # [n_tokens, vocab] of each token's own row
            ids = torch.tensor(text_tokens, device=p.device)
            where = (dist.sort().indices == ids.unsqueeze(1)).nonzero()[:, -1].tolist()
            ranks = [where[k] for k in pick]
        return np.array([tok_p[k] for k in pick]), ranks

    def _commit(self, idx: int, done: np.ndarray, track: np.ndarray, at_end: bool, offset: float, words: List[WordTiming]):
        """:329-357 -- write the last boundary that kept the best token into the word; a boundary found only through
        failed probes may not move the timestamp outwards."""
        if not done[idx] or track[idx, -1] == -1:
            return
        ts = round(offset + (float(track[idx, -1]) / self.sample_rate), 3)
        if track[idx, 0] and not track[idx, 1]:
            if at_end:
                if ts <= words[idx].end:
                    return
            elif ts >= words[idx].start:
                return
        if at_end:
            words[idx].end = ts
        else:
            words[idx].start = ts

    # ---------------------------------------------------------------------------------------------- one step
    def _refine(self, result: WhisperResult, step: str):
        total_duration = round(self._audio.shape[-1] / self.sample_rate, 3)
        at_end = step == "e"
        for words, g_lo, g_hi, edge in self._groups(result, total_duration):
            offset = g_lo[0]
            a, b = round(offset * self.sample_rate), round(g_hi[-1] * self.sample_rate)
            clean = self._audio[a:b + 1].unsqueeze(0)
            max_start = self._samples([w.end for w in words], offset)
            min_end = self._samples([w.start for w in words], offset)
            min_start = self._samples(g_lo, offset)
            max_end = self._samples(g_hi, offset)
            mid_start = min_start + ((max_start - min_start) / 2).round().astype(np.int32)
            mid_end = min_end + ((max_end - min_end) / 2).round().astype(np.int32)
            text_tokens = [t for w in words for t in w.tokens]
            word_tokens = [list(w.tokens) for w in words]
            probe = clean.clone().repeat_interleave(2, 0)               # copy 0: even words, copy 1: odd words
            done = np.less([w.probability for w in words], self.prob_threshold)
            done = np.logical_or(done, [w.duration == 0 for w in words])
            if not self.word_level:
                done[edge != (2 if at_end else 1)] = True
            rows: List[int] = []
            for idx, cut in enumerate(max_start if at_end else min_end):
                row = idx % 2
                rows.extend([row] * len(words[idx].tokens))
                if done[idx]:
                    continue
                if at_end:                                               # mute from the word's end to the next word
                    stop = probe.size(-1) if idx == len(words) - 1 else mid_end[idx + 1]
                    probe[row, cut:stop] = 0
                else:                                                    # mute from the previous word up to the start
                    stop = 0 if idx == 0 else mid_start[idx - 1]
                    probe[row, stop:cut] = 0
            ref_p, ref_rank = self._probe(probe, text_tokens, word_tokens, rows, at_end)
            track = np.zeros((ref_p.shape[-1], 3), dtype=np.int32)       # [failed once, passed once, last good boundary]
            track[:, -1] = -1
            first_cut = (mid_end, max_start) if at_end else (min_end, mid_start)
            for idx, (s, e) in enumerate(zip(*first_cut)):
                if not done[idx]:
                    probe[idx % 2, s:e] = 0
            prev_p = ref_p
            while not np.all(done):
                p, rank = self._probe(probe, text_tokens, word_tokens, rows, at_end)
                abs_drop = ref_p - p
                rel_drop = abs_drop / ref_p
                step_drop = (prev_p - p) / prev_p
                prev_p = p
                for idx in range(len(words)):
                    if done[idx]:
                        continue
                    if at_end:
                        lo, hi, mid = min_end[idx], max_end[idx], mid_end[idx]
                    else:
                        lo, hi, mid = min_start[idx], max_start[idx], mid_start[idx]
                    row = rows[idx]
                    lost_rank = ref_rank[idx] > rank[idx]
                    failed = (abs_drop[idx] > self.abs_prob_decrease or rel_drop[idx] > self.rel_prob_decrease or
                              (self.rel_rel_prob_decrease is not None and step_drop[idx] > self.rel_rel_prob_decrease) or
                              p[idx] < self.prob_threshold or lost_rank)
                    if failed:
                        track[idx][0] = 1
                        if at_end:
                            lo = mid
                        else:
                            hi = mid
                    else:
                        track[idx][1] = 1
                        if at_end:
                            hi = mid
                        else:
                            lo = mid
                    half = round((hi - lo) / 2)
                    if half < self.sample_precision:
                        done[idx] = True
                        self._commit(idx, done, track, at_end, offset, words)
                        continue
                    new_mid = lo + half
                    if failed:                                           # give audio back
                        if at_end:
                            probe[row, lo:new_mid] = clean[0, lo:new_mid]
                        else:
                            probe[row, new_mid:hi] = clean[0, new_mid:hi]
                    elif at_end:                                         # mute more
                        probe[row, new_mid:hi] = 0
                    else:
                        probe[row, lo:new_mid] = 0
                    if at_end:
                        min_end[idx], max_end[idx], mid_end[idx] = lo, hi, new_mid
                    else:
                        min_start[idx], max_start[idx], mid_start[idx] = lo, hi, new_mid
                    if not lost_rank:
                        track[idx][-1] = new_mid
                    ref_p[idx] = p[idx]
