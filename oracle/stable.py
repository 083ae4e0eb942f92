"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the reference's OWN glue for the hot path, on top of oracle/whisper (the restated
openai-whisper), so that the oracle also runs where /root/reference does not exist (GPU box tests,
bench.py's cpu_baseline leg).  Each function cites the reference lines it follows.  When /root/reference IS
importable (this container), tests/golden/make_golden.py runs the reference's real code on oracle/whisper and
tests/test_oracle_glue.py checks that this restatement gives the same words/timestamps.
"""
import string
from dataclasses import dataclass, replace
from itertools import chain
from typing import Callable, List, Optional

import numpy as np
import torch

from .whisper.audio import N_SAMPLES_PER_TOKEN, TOKENS_PER_SECOND
from .whisper.decoding import DecodingOptions, DecodingTask
from .whisper.model import disable_sdpa
from .whisper.timing import dtw, median_filter, merge_punctuations


# ------------------------------------------------------------------------------------------------ decode.py
class DecodingTaskStable(DecodingTask):
    """stable_whisper/decode.py:20-65: encoder output reuse + timestamp-token suppression + nan_to_num."""

    def __init__(self, *args, ts_token_mask=None, audio_features=None, **kwargs):
        self.ts_token_mask = ts_token_mask          # decode.py:23
        self.audio_features = audio_features        # decode.py:24
        super().__init__(*args, **kwargs)

    def _get_audio_features(self, mel):
        if self.audio_features is None:             # decode.py:27-30
            self.audio_features = super()._get_audio_features(mel)
        return self.audio_features

    def _main_loop(self, audio_features, tokens):   # decode.py:33-65
        n_batch = tokens.shape[0]
        sum_logprobs = torch.zeros(n_batch, device=audio_features.device)
        no_speech_probs = [np.nan] * n_batch
        try:
            for i in range(self.sample_len):
                logits = self.inference.logits(tokens, audio_features)
                if i == 0 and self.tokenizer.no_speech is not None:
                    at_sot = logits[:, self.sot_index].float().softmax(dim=-1)
                    no_speech_probs = at_sot[:, self.tokenizer.no_speech].tolist()
                logits = logits[:, -1]
                for f in self.logit_filters:
                    f.apply(logits, tokens)
                if self.ts_token_mask is not None:  # decode.py:14-16, 54
                    logits[:, self.tokenizer.timestamp_begin:][:, self.ts_token_mask] = -np.inf
                logits.nan_to_num_(-np.inf)         # decode.py:56
                tokens, completed = self.decoder.update(tokens, logits, sum_logprobs)
                if completed or tokens.shape[-1] > self.n_ctx:
                    break
        finally:
            self.inference.cleanup_caching()
        return tokens, sum_logprobs, no_speech_probs


class _MinTokens:
    """Benchmark-only logit filter: EOT is suppressed until `n` tokens were sampled (random weights have no
    meaningful EOT).  Mirrors swx_decode_cfg.min_tokens so that oracle and GPU decode the same fixed budget."""

    def __init__(self, eot, sample_begin, n):
        self.eot, self.sample_begin, self.n = eot, sample_begin, n

    def apply(self, logits, tokens):
        if tokens.shape[1] - self.sample_begin < self.n:
            logits[:, self.eot] = -np.inf


@torch.no_grad()
def decode_stable(model, mel, options: DecodingOptions = None, ts_token_mask=None, audio_features=None,
                  min_tokens: int = 0, **kwargs):
    """decode.py:70-110.  Returns (DecodingResult | list, audio_features)."""
    single = mel.ndim == 2
    if single:
        mel = mel.unsqueeze(0)
    options = options or DecodingOptions()
    if kwargs:
        options = replace(options, **kwargs)
    task = DecodingTaskStable(model, options, ts_token_mask=ts_token_mask, audio_features=audio_features)
    if min_tokens:
        # placed right after SuppressBlank/SuppressTokens, before ApplyTimestampRules (same place as the kernel)
        pos = len(task.logit_filters) - (0 if options.without_timestamps else 1)
        task.logit_filters.insert(pos, _MinTokens(task.tokenizer.eot, task.sample_begin, min_tokens))
    result = task.run(mel)
    return (result[0] if single else result), task.audio_features


# ------------------------------------------------------------------------------------------------ timing.py
@dataclass
class WordTiming:          # timing.py:22-28
    word: str
    tokens: List[int]
    start: float
    end: float
    probability: float


def compute_qks(model, tokenizer, text_tokens, mel, tokens, cache):
    """timing.py:41-67: teacher-forced pass with hooks on every cross-attention; token probabilities."""
    cache["qks"] = [None] * model.dims.n_text_layer
    hooks = [blk.cross_attn.register_forward_hook(lambda _, i, o, k=k: cache["qks"].__setitem__(k, o[-1]))
             for k, blk in enumerate(model.decoder.blocks)]
    with torch.no_grad(), disable_sdpa():
        if cache["audio_features"] is None:
            cache["audio_features"] = model.encoder(mel.unsqueeze(0))
        logits = model.decoder(tokens.unsqueeze(0), cache["audio_features"])[0]
        sampled = logits[len(tokenizer.sot_sequence):, : tokenizer.eot]
        probs = sampled.softmax(dim=-1)
        cache["text_token_probs"] = probs[np.arange(len(text_tokens)), text_tokens].tolist()
    for h in hooks:
        h.remove()


def compute_atten_weights(model, tokenizer, text_tokens, mel, num_samples, tokens, cache, medfilt_width=7, qk_scale=1.0):
    """timing.py:70-112, legacy aligner with the model's alignment heads (no dynamic heads)."""
    if cache["qks"] is None:
        compute_qks(model, tokenizer, text_tokens, mel, tokens, cache)
    qks = cache["qks"]
    w = torch.cat([qks[l][:, h] for l, h in model.alignment_heads.indices().T], dim=0)
    w = w[:, len(tokenizer.sot_sequence): -1, : round(num_samples / N_SAMPLES_PER_TOKEN)]
    w = (w * qk_scale).softmax(dim=-1)
    std, mean = torch.std_mean(w, dim=-2, keepdim=True, unbiased=False)
    w = (w - mean) / std
    return median_filter(w, medfilt_width)


def compute_jump_indices(model, cache, **kw):
    """timing.py:166-198 (legacy branch)."""
    weights = compute_atten_weights(model, cache=cache, **kw)
    matrix = weights.mean(dim=0)
    cache["neg_matrix"] = -matrix
    text_indices, time_indices = dtw(-matrix)
    cache["dtw_path"] = (text_indices, time_indices)
    jumps = np.pad(np.diff(text_indices), (1, 0), constant_values=1).astype(bool)
    cache["jump_indices"] = time_indices[jumps].clip(min=0)


def find_alignment(model, tokenizer, text_tokens, mel, num_samples, *, medfilt_width=7, qk_scale=1.0,
                   token_split=None, audio_features=None, return_cache=False):
    """timing.py:202-306 (legacy aligner, no dynamic heads / extra models)."""
    tokens = torch.tensor([*tokenizer.sot_sequence, tokenizer.no_timestamps, *text_tokens, tokenizer.eot]).to(model.device)
    if token_split is None:
        words, word_tokens = tokenizer.split_to_word_tokens(text_tokens + [tokenizer.eot])
    else:
        words, word_tokens = token_split
        words.append(tokenizer.decode([tokenizer.eot]))
        word_tokens.append([tokenizer.eot])
    bounds = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    cache = dict(audio_features=audio_features, jump_indices=None, text_token_probs=None, qks=None)
    compute_jump_indices(model, cache, tokenizer=tokenizer, text_tokens=text_tokens, mel=mel, num_samples=num_samples,
                         tokens=tokens, qk_scale=qk_scale, medfilt_width=medfilt_width)
    jump_times = cache["jump_indices"] / TOKENS_PER_SECOND
    starts = jump_times[bounds[:-1]]
    ends = jump_times[bounds[1:]]
    probs = [np.mean(cache["text_token_probs"][i:j]) for i, j in zip(bounds[:-1], bounds[1:])]
    out = [WordTiming(w, t, s, e, p) for w, t, s, e, p in zip(words, word_tokens, starts, ends, probs)]
    return (out, cache) if return_cache else out


def split_tokens(tokens: List[int], tokenizer):
    """timing.py:309-341."""
    by_space = getattr(tokenizer, "language_code", tokenizer.language) not in {"zh", "ja", "th", "lo", "my"}
    text = tokenizer.decode_with_timestamps(tokens)
    words, word_tokens, curr = [], [], []
    is_append = False
    curr_text = ""
    for token in tokens:
        curr.append(token)
        curr_text = tokenizer.decode(curr)
        whole = token >= tokenizer.eot
        if not whole:
            whole = text[:len(curr_text)] == curr_text
            if whole and by_space:
                is_append = not (curr_text.startswith(" ") or curr_text.strip() in string.punctuation)
        if whole:
            if is_append and len(words) != 0:
                words[-1] += curr_text
                word_tokens[-1].extend(curr)
            else:
                words.append(curr_text)
                word_tokens.append(curr)
            text = text[len(curr_text):]
            curr = []
    if len(curr) != 0:
        words.append(curr_text if len(text) == 0 else text)
        word_tokens.append(curr)
    elif len(text) != 0:
        words[-1] += text
    return words, word_tokens


def split_word_tokens(segments, tokenizer, *, padding=None, split_callback: Callable = None, pad_first_seg=True):
    """timing.py:344-392 (char_split=False)."""
    if padding is not None:
        padding = tokenizer.encode(padding) if isinstance(padding, str) else [padding]
    tokens, seg_indices, words, word_tokens = [], [], [], []
    for i, s in enumerate(segments):
        temp = [t for t in s["tokens"] if not isinstance(t, int) or t < tokenizer.eot]
        cw, cwt = split_tokens(temp, tokenizer) if split_callback is None else split_callback(temp, tokenizer)
        assert len(cw) == len(cwt)
        if (padding is not None and cwt[0][0] != padding and (len(tokens) == 0 or tokens[-1] != padding)
                and (pad_first_seg or i != 0)):
            tokens.extend(padding)
            words.append(None)
            word_tokens.append(padding)
        seg_indices.extend([i] * len(cw))
        tokens.extend(list(chain.from_iterable(cwt)))
        words.extend(cw)
        word_tokens.extend(cwt)
    return tokens, (words, word_tokens), seg_indices


def pop_empty_alignment(alignment, seg_indices=None):
    """timing.py:395-407."""
    if seg_indices is not None:
        pos = len(seg_indices)
        empty = {}
        for i in reversed(range(len(alignment))):
            assert pos != -1
            if alignment[i].word is None:
                empty[seg_indices[pos]] = alignment.pop(i)
            else:
                pos -= 1
        return empty
    return list(reversed([alignment.pop(i) for i in reversed(range(len(alignment))) if alignment[i].word is None]))


def add_word_timestamps(*, segments, model, tokenizer, mel, num_samples,
                        prepend_punctuations="\"'“¿([{-", append_punctuations="\"'.。,，!！?？:：”)]}、",
                        audio_features=None, min_word_dur=0.1, split_callback=None, gap_padding=" ...",
                        pad_first_seg=True, **kwargs):
    """timing.py:411-500 (legacy aligner)."""
    if len(segments) == 0:
        return
    min_word_dur = min_word_dur or 0
    for seg in segments:
        seg["words"] = []
    text_tokens, token_split, seg_indices = split_word_tokens(segments, tokenizer, padding=gap_padding,
                                                              split_callback=split_callback, pad_first_seg=pad_first_seg)
    alignment = find_alignment(model, tokenizer, text_tokens, mel, num_samples, **kwargs, token_split=token_split,
                               audio_features=audio_features)
    alt_begin = pop_empty_alignment(alignment, seg_indices)
    merge_punctuations(alignment, prepend_punctuations, append_punctuations)
    offset = segments[0]["seek"]
    assert len(alignment) == len(seg_indices)
    for i, timing in zip(seg_indices, alignment):
        if len(timing.tokens) != 0:
            start, end = timing.start, timing.end
            if len(segments[i]["words"]) == 0 and (end - start) < min_word_dur and i in alt_begin:
                start = alt_begin[i].start
            segments[i]["words"].append(dict(word=timing.word, start=round(offset + start, 3), end=round(offset + end, 3),
                                             probability=timing.probability, tokens=timing.tokens))
    for seg in segments:
        if len(seg["words"]) > 0:
            seg["start"] = seg["words"][0]["start"]
            seg["end"] = seg["words"][-1]["end"]
