"""Golden fixtures for ``refine()``'s bisection (stable_ts_amd/refiner.py): the reference's ``Refiner``
(stable_whisper/non_whisper/refinement.py) driven by a deterministic synthetic inference function on seeded audio and
word-timed results; tests/test_refiner_cpu.py feeds the same function to stable_ts_amd.refiner.Refiner.

    python tests/golden/make_refiner_golden.py
"""
import contextlib
import copy
import gzip
import io
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

VOCAB = 48


def make_inference(seed: int, with_vocab: bool, calls: list = None):
    """inference_func(audio[2, n], tokens) -> probabilities [2, T] or [2, T, VOCAB].

    Token j "lives" in the j-th of T equal slices of the segment: its probability is a hashed base confidence times the
    (powered) fraction of that slice that is not muted, so muting into a word lowers it monotonically.  With a vocabulary
    axis a hashed competitor token takes over once the true token has lost enough mass (rank change)."""
    def infer(audio, tokens):
        T = len(tokens)
        n = int(audio.shape[-1])
        if calls is not None:
            calls.append((n, T, int((audio != 0).sum())))
        edges = np.linspace(0, n, T + 1).round().astype(int)
        live = (audio != 0).float()
        cs = torch.cat([torch.zeros(2, 1), live.cumsum(-1)], dim=-1)
        out = torch.zeros(2, T, VOCAB) if with_vocab else torch.zeros(2, T)
        for j, t in enumerate(tokens):
            h = (int(t) * 2654435761 + seed * 131 + j * 7) % 100003
            base = 0.35 + 0.6 * (h % 1000) / 1000
            a, b = int(edges[j]), max(int(edges[j + 1]), int(edges[j]) + 1)
            frac = (cs[:, b] - cs[:, a]) / (b - a)
            p = base * frac.clamp(0, 1) ** (0.5 + (h % 7) / 4)
            if with_vocab:
                true_id = int(t) % VOCAB
                rival = (true_id + 1 + h % (VOCAB - 1)) % VOCAB
                rest = (1 - p)
                out[:, j, :] = (rest * 0.5 / (VOCAB - 2)).unsqueeze(-1)
                out[:, j, rival] = rest * 0.5 * (0.2 + 0.7 * ((h // 7) % 10) / 10)
                out[:, j, true_id] = p
            else:
                out[:, j] = p
        return out
    return infer


def synth_case(seed: int):
    """(audio, result dict, refiner options, with_vocab)"""
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    seconds = rng.choice([20, 45, 80])
    audio = 0.1 * torch.randn(seconds * 16000, generator=g) + 0.2        # never exactly zero
    with_vocab = rng.random() < 0.5
    segs = []
    t = rng.uniform(0.0, 1.0)
    tok = 5
    while t < seconds - 3:
        words = []
        for _ in range(rng.choice([1, 2, 4, 7, 12])):
            d = rng.choice([0.0, 0.12, 0.3, 0.5, 0.9, 2.0])
            gap = rng.choice([0.0, 0.0, 0.05, 0.3, 1.0])
            a = round(t + gap, 3)
            b = round(min(a + d, seconds - 0.01), 3)
            if a >= seconds - 0.5:
                break
            n_tok = rng.choice([1, 1, 2, 3])
            toks = [(tok + k) % VOCAB if with_vocab else tok + k for k in range(n_tok)]
            tok += n_tok
            words.append(dict(word=f" w{tok}", start=a, end=b, probability=round(rng.choice([0.2, 0.55, 0.7, 0.9, 0.99]), 3),
                              tokens=toks))
            t = b
        if words:
            segs.append(dict(start=words[0]["start"], end=words[-1]["end"], text="".join(w["word"] for w in words), words=words))
        t += rng.choice([0.0, 0.4, 2.0])
    opts = dict(steps=rng.choice(["se", "s", "e", "es"]),
                rel_prob_decrease=rng.choice([0.03, 0.1, 0.3]),
                abs_prob_decrease=rng.choice([0.05, 0.15]),
                rel_rel_prob_decrease=rng.choice([None, None, 0.1]),
                prob_threshold=rng.choice([0.5, 0.3]),
                rel_dur_change=rng.choice([0.5, 0.5, None, 1.0]),
                abs_dur_change=rng.choice([None, None, 0.4]),
                word_level=rng.random() < 0.8,
                precision=rng.choice([None, 0.05, 0.2]),
                max_inference_tokens=rng.choice([100, 20, 8]))
    return audio, dict(language="en", segments=segs), opts, with_vocab


def snapshot(res):
    return [[[w.word, w.start, w.end] for w in s.words] for s in res.segments]


def run(refiner_cls, result_cls, seed: int, extra=None):
    audio, rd, opts, with_vocab = synth_case(seed)
    calls = []
    res = result_cls(copy.deepcopy(rd))
    rf = refiner_cls(make_inference(seed, with_vocab, calls), **opts, **(extra or {}))
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter("ignore")
        try:
            out = rf.refine(audio, res)
        except Exception as e:
            return dict(error=type(e).__name__), calls
    return snapshot(out), calls


def main():
    from make_golden import import_reference
    sw = import_reference()
    from stable_whisper.non_whisper.refinement import Refiner as RefRefiner
    cases = {}
    for seed in range(30):
        snap, calls = run(RefRefiner, sw.WhisperResult, seed, extra=dict(verbose=None))
        cases[str(seed)] = dict(out=snap, n_calls=len(calls))
    out = os.path.join(HERE, "refiner_cases.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(cases, separators=(",", ":")).encode("utf-8"))
    n_err = sum(1 for c in cases.values() if isinstance(c["out"], dict))
    print(f"wrote {len(cases)} cases ({n_err} raising, {sum(c['n_calls'] for c in cases.values())} inference calls) -> {out} "
          f"({os.path.getsize(out) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
