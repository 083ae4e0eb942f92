"""Golden fixtures for the forced-alignment window state machine (stable_ts_amd/aligner.py).

The reference's ``Aligner`` (stable_whisper/non_whisper/alignment.py) is generic over ``inference_func``: this script
runs it with a deterministic synthetic inference function (`make_inference`: words are laid onto the loud 20-ms units of
the segment, durations / confidences hashed from the token ids, occasionally zero-length, over-long or split into two
pieces) on seeded synthetic audio and text, and stores the resulting WhisperResult per case.
tests/test_aligner_cpu.py feeds the same function, audio and text to stable_ts_amd.aligner.Aligner and requires
identical results; where /root/reference is present it also does so live on further seeds.

    python tests/golden/make_aligner_golden.py
"""
import gzip
import json
import os
import random
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

UNIT = 320          # samples per 20-ms unit


def make_inference(seed: int, calls: list = None):
    """inference_func(audio_segment, word_tokens) -> list of word dicts (times relative to the segment)."""
    def infer(audio_segment, word_tokens):
        x = audio_segment.detach().float().abs()
        n_units = int(x.shape[-1]) // UNIT
        loud = (x[: n_units * UNIT].reshape(-1, UNIT).mean(1) > 0.01).numpy() if n_units else np.zeros(0, bool)
        if calls is not None:
            calls.append((int(x.shape[-1]), [w.word for w in word_tokens]))
        out = []
        t = 0
        for w in word_tokens:
            h = (sum(int(v) for v in w.tokens) * 2654435761 + seed * 97 + len(w.word)) % 100003
            if w.is_padding:
                need = 2 + h % 5
            elif h % 17 == 0:
                need = 0                     # "instant" word
            elif h % 13 == 0:
                need = 150 + h % 60          # implausibly long (3-4 s)
            else:
                need = 5 + h % 25            # 0.1-0.6 s
            while t < n_units and not loud[t]:
                t += 1
            a, b = t, min(t + need, n_units)
            t = b
            prob = 0.05 + 0.9 * ((h * 7) % 100) / 100
            if h % 11 == 0 and len(w.word) > 3 and len(w.tokens) > 1 and not w.is_padding:
                k = len(w.word) // 2         # the model timed the word as two pieces
                m = (a + b) // 2
                out.append(dict(word=w.word[:k], start=a * 0.02, end=m * 0.02, probability=prob, tokens=w.tokens[:1]))
                out.append(dict(word=w.word[k:], start=m * 0.02, end=b * 0.02, probability=prob / 2, tokens=w.tokens[1:]))
            else:
                out.append(dict(word=w.word, start=a * 0.02, end=b * 0.02, probability=prob, tokens=list(w.tokens)))
        return out
    return infer


def synth_case(seed: int):
    """(audio f32[n], text, options) for one case."""
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    seconds = rng.choice([35, 70, 120, 190])
    n = seconds * 16000
    audio = 0.1 * torch.randn(n, generator=g)
    t = rng.uniform(1.0, 6.0)
    while t < seconds:                         # silent stretches: short pauses and a few >= 5 s
        d = rng.choice([0.2, 0.4, 0.8, 1.5, 3.0, 5.5, 7.0, 12.0])
        a, b = int(t * 16000), min(int((t + d) * 16000), n)
        audio[a:b] = 0.0 if rng.random() < 0.7 else 0.0005 * torch.randn(b - a, generator=g)
        t += d + rng.uniform(2.0, 25.0)
    n_words = rng.choice([15, 60, 150, 320])
    ids = []
    for _ in range(n_words):
        ids.append(rng.choice([19, 20, 22, 23, 25, 26]) + 3 * rng.randrange(0, 4000))      # starts a word (id % 3 != 0)
        if rng.random() < 0.3:
            ids.append(18 + 3 * rng.randrange(0, 4000))                                     # continuation piece
        r = rng.random()
        if r < 0.12:
            ids.append(0)                      # '.'
        elif r < 0.22:
            ids.append(1)                      # ','
        elif r < 0.26:
            ids.append(3)                      # '?'
        elif r < 0.28:
            ids.extend([16, 8])                # ' ' + '(' : a free-standing opening bracket
        elif r < 0.30:
            ids.append(9)                      # ')'
    opts = dict(token_step=rng.choice([20, 50, 100, 100, 200]),
                word_dur_factor=rng.choice([2.0, 2.0, None, 1.2]),
                max_word_dur=rng.choice([3.0, 3.0, None, 1.0]),
                nonspeech_skip=rng.choice([5.0, 5.0, None, 3.0]),
                fast_mode=rng.random() < 0.2,
                failure_threshold=rng.choice([None, None, 0.3]),
                remove_instant_words=rng.random() < 0.2,
                original_split=rng.random() < 0.25,
                suppress_silence=rng.random() < 0.8,
                presplit=rng.choice([True, True, False, [".", "?"]]),
                regroup=rng.choice([True, True, False]),
                gap_padding=rng.choice([" ...", " ...", None]),
                min_word_dur=rng.choice([None, None, 0.2]),
                q_levels=rng.choice([20, 20, 10]), k_size=rng.choice([5, 5, 3]),
                nonspeech_error=rng.choice([0.1, 0.1, 0.3]),
                use_word_position=rng.random() < 0.8, suppress_word_ts=rng.random() < 0.8)
    return audio, ids, opts


def case_text(ids, tok, original_split: bool, seed: int) -> str:
    text = tok.decode(ids)
    if original_split:                         # caller-provided line breaks
        rng = random.Random(seed + 1)
        parts = text.split(" ")
        for i in range(3, len(parts), rng.choice([5, 9, 14])):
            parts[i] = "\n" + parts[i]
        text = " ".join(parts)
    return text


def snapshot(res) -> dict:
    if res is None:
        return dict(none=True)
    segs = [[[w.word, w.start, w.end, round(float(w.probability), 12), list(w.tokens)] for w in s.words] for s in res.segments]
    return dict(segments=segs, nonspeech=[[d["start"], d["end"]] for d in res.nonspeech_sections],
                history=res.regroup_history)


def run(aligner_cls, seed: int, tok, extra=None):
    audio, ids, opts = synth_case(seed)
    text = case_text(ids, tok, opts["original_split"], seed)
    calls = []
    al = aligner_cls(inference_func=make_inference(seed, calls), decode=tok.decode, encode=tok.encode, **opts, **(extra or {}))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            res = al.align(audio, text)
        except Exception as e:                 # contract violations surface as the same exception class on both sides
            return dict(error=type(e).__name__), calls
    return snapshot(res), calls


def main():
    from make_golden import import_reference
    import_reference()
    from stable_whisper.non_whisper.alignment import Aligner as RefAligner
    from stable_ts_amd.tokenizer import get_tokenizer
    tok = get_tokenizer(False, num_languages=99)
    cases = {}
    for seed in range(40):
        snap, calls = run(RefAligner, seed, tok, extra=dict(verbose=None))
        cases[str(seed)] = dict(out=snap, n_calls=len(calls))
    out = os.path.join(HERE, "aligner_cases.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(cases, ensure_ascii=False, separators=(",", ":")).encode("utf-8"))
    n_err = sum(1 for c in cases.values() if "error" in c["out"])
    print(f"wrote {len(cases)} cases ({n_err} raising) -> {out} ({os.path.getsize(out) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
