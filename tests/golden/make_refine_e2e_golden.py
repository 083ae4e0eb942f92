"""Golden for model.refine() END TO END (stable_whisper/alignment.py:512-635 + non_whisper/refinement.py): the reference's
own refine() on the CPU oracle model, applied to the reference's own align() result of the `align_tiny_en` case of
reference_glue.json.  Also the temperature-ladder golden: the reference's transcribe() with a ladder whose T = 0 attempt
must be rejected (thresholds chosen so), recording the attempt decisions that do not depend on the sampling stream.

    python tests/golden/make_refine_e2e_golden.py      (this container only: needs /root/reference)
"""
import json
import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402


def words_of(res):
    return [dict(word=w.word, start=float(w.start), end=float(w.end), probability=float(w.probability),
                 tokens=[int(t) for t in w.tokens]) for w in res.all_words()]


def run():
    sw = G.import_reference()
    from oracle.whisper.model import build_model
    glue = json.load(open(os.path.join(HERE, "reference_glue.json")))
    g = glue["align_tiny_en"]
    c = g["case"]
    model = build_model(c["model"], seed=1234, std=0.02, embed_gain=c["gain"], ts_gain=c["ts_gain"])
    sw.modify_model(model)
    audio = G.synth_audio(c["seconds"], c["seed"])
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = model.align(audio, g["text"], language="en", verbose=None, ignore_compatibility=True, regroup=False,
                          suppress_silence=False, original_split=False)
        before = words_of(res)
        # random weights give word probabilities of ~1e-5: the default prob_threshold (0.5) would stop every search at once
        for name, kw in [("default_thresholds", dict()),
                         ("both_ends", dict(prob_threshold=0.0, precision=0.1)),
                         ("coarse_rel", dict(steps="se", precision=0.2, rel_prob_decrease=0.1, prob_threshold=0.0)),
                         ("starts_only", dict(steps="s", precision=0.1, prob_threshold=0.0, rel_rel_prob_decrease=0.5))]:
            r = sw.WhisperResult(res.to_dict())
            r = model.refine(audio, r, verbose=None, **kw)
            out[name] = dict(kw=kw, words=words_of(r))
            moved = sum(abs(a["start"] - b["start"]) > 1e-9 or abs(a["end"] - b["end"]) > 1e-9 for a, b in zip(before, out[name]["words"]))
            print("refine", name, len(out[name]["words"]), "words,", moved, "moved")
    with open(os.path.join(HERE, "reference_refine_e2e.json"), "w") as f:
        json.dump(dict(case=c, text=g["text"], before=before, refined=out), f, indent=0)


if __name__ == "__main__":
    run()
