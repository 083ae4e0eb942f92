"""The forced-alignment window state machine (stable_ts_amd/aligner.py) against the reference's ``Aligner``
(stable_whisper/non_whisper/alignment.py), both driven by the SAME synthetic inference function.

* golden: tests/golden/aligner_cases.json.gz = the reference's results on 40 seeded cases
  (tests/golden/make_aligner_golden.py): every word, its start/end/probability/tokens, the segmentation, the detected
  non-speech sections and the number of inference calls must be identical (host control flow on ms-rounded floats: exact).
* live: where /root/reference is importable, more seeds are compared live, including the per-window inference inputs.
"""
import gzip
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import make_aligner_golden as mg  # noqa: E402

from stable_ts_amd.aligner import Aligner, merge_punctuations, tokens_to_word_tokens, WordToken  # noqa: E402
from stable_ts_amd.tokenizer import get_tokenizer  # noqa: E402


def _tok():
    return get_tokenizer(False, num_languages=99)


def _norm(x):
    return json.loads(json.dumps(x))


def test_aligner_matches_reference_golden():
    with gzip.open(os.path.join(HERE, "golden", "aligner_cases.json.gz"), "rb") as f:
        cases = json.loads(f.read().decode("utf-8"))
    assert len(cases) == 40
    tok = _tok()
    n_multi = 0
    for seed, want in cases.items():
        got, calls = mg.run(Aligner, int(seed), tok)
        assert _norm(got) == want["out"], (seed, mg.synth_case(int(seed))[2])
        assert len(calls) == want["n_calls"], seed
        n_multi += len(calls) > 2
    assert n_multi >= 20           # most cases need several windows / re-alignments


def test_word_token_grouping():
    tok = _tok()
    ids = [20, 3, 25, 0, 30, 31, 16, 8, 40, 9, 1, 43]         # " aaau? aaaz.aabe aabf (" + " aabo),  aabr"
    words = tokens_to_word_tokens(ids, tok.decode, True)
    assert "".join(w.word for w in words) == tok.decode(ids)
    assert [t for w in words for t in w.tokens] == ids
    assert all(w.word.strip() not in ("?", ".", ",", "(") for w in words)   # single marks were merged into a neighbour
    # a free-standing opening bracket joins the NEXT word, a closing mark the previous one; right to left, so "." first
    # joins ")" and the pair ")." (no longer a single mark) stays a word of its own -- the reference's behaviour
    ws = [WordToken(" a", [1]), WordToken(" (", [2]), WordToken(" b", [3]), WordToken(")", [4]), WordToken(".", [5])]
    merge_punctuations(ws)
    assert [w.word for w in ws] == [" a", " ( b", ")."] and [w.tokens for w in ws] == [[1], [2, 3], [4, 5]]


def test_aligner_argument_errors():
    tok = _tok()
    with pytest.raises(ValueError):
        Aligner(lambda a, w: [], tok.decode, tok.encode, failure_threshold=1.5)
    with pytest.raises(TypeError):
        Aligner(lambda a, w: [], tok.decode, tok.encode, not_an_option=1)
    with pytest.raises(NotImplementedError):
        Aligner(lambda a, w: [], tok.decode, tok.encode, vad=True)
    import torch
    al = Aligner(lambda a, w: [], tok.decode, tok.encode)       # fewer output words than requested: contract violation
    with pytest.raises(RuntimeError):
        al.align(0.1 * torch.randn(16000 * 3), " aaat aaau")


@pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")
def test_aligner_matches_reference_live():
    from make_golden import import_reference
    import_reference()
    from stable_whisper.non_whisper.alignment import Aligner as RefAligner
    tok = _tok()
    for seed in range(1000, 1018):
        want, ref_calls = mg.run(RefAligner, seed, tok, extra=dict(verbose=None))
        got, calls = mg.run(Aligner, seed, tok)
        assert calls == ref_calls, (seed, mg.synth_case(seed)[2])         # same windows, same word batches
        assert _norm(got) == _norm(want), (seed, mg.synth_case(seed)[2])


@pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")
def test_aligner_reproduces_reference_align_with_whisper_inference():
    """The committed golden `align_tiny_en` (tests/golden/reference_glue.json) is the reference's ``model.align`` on the
    CPU oracle model.  Here the reference's own seam-B2 callable (get_whisper_alignment_func on that oracle model) is
    plugged into stable_ts_amd's Aligner: words, times and probabilities must equal the golden exactly."""
    from types import SimpleNamespace
    import make_golden as G
    sw = G.import_reference()
    from oracle.whisper.model import build_model
    from oracle.whisper.tokenizer import get_tokenizer as oracle_tokenizer
    from stable_whisper.alignment import get_whisper_alignment_func
    with open(os.path.join(HERE, "golden", "reference_glue.json")) as f:
        g = json.load(f)["align_tiny_en"]
    c = g["case"]
    model = build_model(c["model"], seed=1234, std=0.02, embed_gain=c["gain"], ts_gain=c["ts_gain"])
    sw.modify_model(model)
    tok = oracle_tokenizer(False, num_languages=model.num_languages)
    opts = SimpleNamespace(align=SimpleNamespace(extra_models=None, dynamic_heads=None, aligner="legacy"))
    func = get_whisper_alignment_func(model, tok, None, opts)
    al = Aligner(func, tok.decode, tok.encode, regroup=False, suppress_silence=False)
    res = al.align(G.synth_audio(c["seconds"], c["seed"]), g["text"])
    got = [(w.word, w.start, w.end, list(w.tokens)) for w in res.all_words()]
    want = [(w["word"], w["start"], w["end"], w["tokens"]) for w in g["words"]]
    assert got == want
    for a, b in zip(res.all_words(), g["words"]):
        assert abs(a.probability - b["probability"]) < 1e-9


def _align_words_case(seed: int, tok):
    import random
    rng = random.Random(seed)
    audio, ids, opts = mg.synth_case(seed)
    seconds = audio.shape[-1] / 16000
    segs, t, k = [], rng.uniform(0.0, 2.0), 0
    while t < seconds - 2 and k < len(ids):
        n = rng.choice([1, 3, 6, 10])
        piece = ids[k:k + n]
        k += n
        d = rng.choice([0.0, 0.6, 2.5, 6.0, 11.0])
        a, b = round(t, 3), round(min(t + d, seconds), 3)
        segs.append(dict(start=a, end=b, text=tok.decode(piece).strip() if rng.random() < 0.5 else tok.decode(piece)))
        t = b + rng.choice([0.0, 0.3, 2.0])
    keep = {k_: opts[k_] for k_ in ("suppress_silence", "regroup")}
    return audio, segs, keep


def _snap_segments(res):
    return [[s.start, s.end, s.text, None if s.words is None else
             [[w.word, w.start, w.end, round(float(w.probability), 12), list(w.tokens)] for w in s.words]]
            for s in res.segments]


@pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")
def test_align_words_matches_reference_live():
    import contextlib
    import copy
    import io
    import warnings
    from make_golden import import_reference
    import_reference()
    from stable_whisper.non_whisper.alignment import Aligner as RefAligner
    tok = _tok()
    for seed in range(2000, 2015):
        audio, segs, keep = _align_words_case(seed, tok)
        outs = []
        for cls, extra in ((RefAligner, dict(verbose=None)), (Aligner, {})):
            al = cls(inference_func=mg.make_inference(seed), decode=tok.decode, encode=tok.encode, token_step=448, **keep, **extra)
            with warnings.catch_warnings(), contextlib.redirect_stderr(io.StringIO()):
                warnings.simplefilter("ignore")
                outs.append(_norm(_snap_segments(al.align_words(audio, copy.deepcopy(segs)))))
        assert outs[0] == outs[1], (seed, keep)
        # batched inference (segments are independent) gives the same result as one call per segment
        f = mg.make_inference(seed)
        al = Aligner(inference_func=f, decode=tok.decode, encode=tok.encode, token_step=448, **keep)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = al.align_words(audio, copy.deepcopy(segs), batch_size=4,
                                 batch_inference=lambda chunks, words: [f(c, w) for c, w in zip(chunks, words)])
        assert _norm(_snap_segments(res)) == outs[0]


@pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")
def test_align_wrapper_end_to_end_on_cpu(monkeypatch):
    """stable_ts_amd.alignment.align / align_words / refine wrappers (argument handling, tokenizer and language
    plumbing, option pass-through) with the device callables replaced by the reference's own seam callables on the CPU
    oracle model: align() must reproduce the committed golden of the reference's model.align exactly."""
    from types import SimpleNamespace
    import make_golden as G
    import stable_ts_amd.alignment as A
    sw = G.import_reference()
    from oracle.whisper.model import build_model
    from stable_whisper.alignment import get_whisper_alignment_func, get_whisper_refinement_func
    with open(os.path.join(HERE, "golden", "reference_glue.json")) as f:
        g = json.load(f)["align_tiny_en"]
    c = g["case"]
    model = build_model(c["model"], seed=1234, std=0.02, embed_gain=c["gain"], ts_gain=c["ts_gain"])
    sw.modify_model(model)
    opts = SimpleNamespace(align=SimpleNamespace(extra_models=None, dynamic_heads=None, aligner="legacy"))

    def fake_alignment_func(m, tokenizer):
        f = get_whisper_alignment_func(m, tokenizer, None, opts)
        f.batch = lambda chunks, words: [f(ch, w) for ch, w in zip(chunks, words)]
        return f

    monkeypatch.setattr(A, "make_alignment_func", fake_alignment_func)
    monkeypatch.setattr(A, "make_refinement_func", lambda m, tokenizer: get_whisper_refinement_func(m, tokenizer, None, False))
    audio = G.synth_audio(c["seconds"], c["seed"])
    res = A.align(model, audio, g["text"], language="en", regroup=False, suppress_silence=False)
    assert [(w.word, w.start, w.end, list(w.tokens)) for w in res.all_words()] == \
        [(w["word"], w["start"], w["end"], w["tokens"]) for w in g["words"]]
    assert res.language == "en"
    res2 = A.align(model, audio, g["text"], language="en")                       # defaults: silence suppression + regroup
    assert "".join(w.word for w in res2.all_words()) == "".join(w["word"] for w in g["words"])
    assert res2.regroup_history.startswith("isp=1_cm=")
    with pytest.raises(ValueError):
        A.align(model, audio, g["text"], language="en", token_step=10 ** 6)
    with pytest.raises(TypeError):
        A.align(model, audio, g["text"], language="en", no_such_option=1)
    # align_words keeps the segmentation it is given and re-times the words inside each segment
    segs = [dict(start=s.start, end=s.end, text=s.text) for s in res2.segments]
    res3 = A.align_words(model, audio, segs, language="en", regroup=False)
    assert [s.text for s in res3.segments] == [s["text"] for s in segs] and res3.has_words
    assert all(s0["start"] <= w.start <= w.end <= s0["end"] + 1e-9 for s, s0 in zip(res3.segments, segs) for w in s.words)
    # refine runs on the result and only moves starts later / ends earlier within the allowed range
    before = [(w.start, w.end) for w in res.all_words()]
    out = A.refine(model, audio, res, precision=0.5)
    assert out is res and len(out.all_words()) == len(before)
    assert all(w.start <= w.end for w in out.all_words())


@pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")
@pytest.mark.parametrize("text,seconds", [("", 5.0), (" .", 5.0), (" aaat", 0.05), (" aaat aaau. aaax", 0.4),
                                          (" aaat\n\n aaau", 12.0), ("aaat   aaau\taaax", 12.0)])
def test_aligner_degenerate_inputs_match_reference(text, seconds):
    import contextlib
    import io
    import warnings
    import torch
    from make_golden import import_reference
    import_reference()
    from stable_whisper.non_whisper.alignment import Aligner as RefAligner
    tok = _tok()
    audio = 0.1 * torch.randn(int(seconds * 16000), generator=torch.Generator().manual_seed(3))
    outs = []
    for cls, extra in ((RefAligner, dict(verbose=None)), (Aligner, {})):
        al = cls(inference_func=mg.make_inference(7), decode=tok.decode, encode=tok.encode, original_split=True, **extra)
        with warnings.catch_warnings(), contextlib.redirect_stderr(io.StringIO()):
            warnings.simplefilter("ignore")
            try:
                outs.append(_norm(mg.snapshot(al.align(audio, text))))
            except Exception as e:
                outs.append(type(e).__name__)
    assert outs[0] == outs[1], (text, seconds, outs)


@pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")
@pytest.mark.parametrize("variant", [dict(), dict(dynamic_heads="4,2"), dict(aligner="new"), dict(extra=True)])
def test_align_on_standin_matches_reference_align(monkeypatch, variant):
    """stable_ts_amd.alignment.align end to end on the CPU stand-in (its own seam-B2 callable, incl. the head-selection
    variants) vs the reference's model.align on the same oracle model"""
    import warnings
    import make_golden as G
    import stable_ts_amd.alignment as A
    sw = G.import_reference()
    from oracle.whisper.model import build_model
    from oracle_engine import CpuWhisper, install
    install(monkeypatch)
    ref = build_model("tiny.en", seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5)
    sw.modify_model(ref)
    mine = CpuWhisper(ref)
    mine.manual_attention_encoder = True          # the reference's align encodes inside disable_sdpa() (timing.py:58-60)
    kw_ref, kw_mine = dict(variant), dict(variant)
    if kw_ref.pop("extra", None):
        kw_mine.pop("extra")
        other = build_model("tiny.en", seed=99, std=0.02, embed_gain=2.0, ts_gain=0.5)
        sw.modify_model(other)
        kw_ref["extra_models"], kw_mine["extra_models"] = [other], [CpuWhisper(other)]
        kw_mine["extra_models"][0].manual_attention_encoder = True
    audio = G.synth_audio(50.0, seed=4)
    text = " aaat aaau. aaax aabc, aaat aaau aaax. aabc aaat? aaau aaax aabc aaat."
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = ref.align(audio, text, language="en", verbose=None, ignore_compatibility=True, **kw_ref)
        got = A.align(mine, audio, text, language="en", **kw_mine)
    snap = lambda r: [(w.word, w.start, w.end, round(float(w.probability), 9), list(w.tokens)) for w in r.all_words()]
    assert snap(got) == snap(want) and len(snap(want)) > 5
    assert [(s.start, s.end, s.text) for s in got.segments] == [(s.start, s.end, s.text) for s in want.segments]
    assert got.to_dict() == want.to_dict()


@pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")
@pytest.mark.parametrize("as_dicts,opts", [(False, dict()), (True, dict(suppress_silence=False, regroup=False)),
                                           (False, dict(inplace=False, min_word_dur=0.2, normalize_text=False))])
def test_align_words_on_standin_matches_reference(monkeypatch, as_dicts, opts):
    """stable_ts_amd.alignment.align_words end to end (batched device callable, segment slicing) vs the reference's
    model.align_words on the same oracle model; align()-only options are refused the same way"""
    import copy
    import warnings
    import make_golden as G
    import stable_ts_amd.alignment as A
    from stable_ts_amd.result import WhisperResult
    sw = G.import_reference()
    import stable_whisper
    from oracle.whisper.model import build_model
    from oracle_engine import CpuWhisper, install
    install(monkeypatch)
    ref = build_model("tiny.en", seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5)
    sw.modify_model(ref)
    mine = CpuWhisper(ref)
    mine.manual_attention_encoder = True
    audio = G.synth_audio(62.0, seed=587)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d = ref.transcribe(audio, language="en", verbose=None, ignore_compatibility=True, temperature=0.0, logprob_threshold=None,
                           compression_ratio_threshold=None, no_speech_threshold=None, sample_len=24, regroup=False).to_dict()
        given = (lambda cls: [dict(start=s["start"], end=s["end"], text=s["text"]) for s in d["segments"]]) if as_dicts else \
            (lambda cls: cls(copy.deepcopy(d)))
        want = ref.align_words(audio, given(stable_whisper.WhisperResult), language="en", verbose=None, ignore_compatibility=True, **opts)
        got = A.align_words(mine, audio, given(WhisperResult), language="en", batch_size=3, **opts)
    snap = lambda r: [(w.word, w.start, w.end, round(float(w.probability), 9), list(w.tokens)) for w in r.all_words()]
    assert snap(got) == snap(want) and len(snap(want)) > 5
    assert [(s.start, s.end, s.text) for s in got.segments] == [(s.start, s.end, s.text) for s in want.segments]
    assert got.to_dict() == want.to_dict()
    for fn, model, cls in ((ref.align_words, None, stable_whisper.WhisperResult), (A.align_words, mine, WhisperResult)):
        with pytest.raises(TypeError, match="unexpected keyword"):
            if model is None:
                fn(audio, given(cls), language="en", ignore_compatibility=True, max_word_dur=1.0)
            else:
                fn(model, audio, given(cls), language="en", max_word_dur=1.0)


def test_sharp_align_fixture_reproduced_by_host_code_on_standin(monkeypatch):
    """tests/golden/reference_glue_sharp.json (the reference's align() / seam B2 on the oracle with SHARP weights; the GPU tests
    hold the f32 AND the fp16 device path against it) is reproduced exactly by this package's host code on the CPU stand-in"""
    import json
    import warnings
    import make_golden as G
    import stable_ts_amd.alignment as A
    from oracle.whisper.model import build_model
    from oracle.whisper.tokenizer import get_tokenizer
    from oracle_engine import CpuWhisper, install
    install(monkeypatch)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_glue_sharp.json")) as f:
        g = json.load(f)["align_base_en_sharp"]
    c = g["case"]
    mine = CpuWhisper(build_model(c["model"], **c["weights"]))
    mine.manual_attention_encoder = True
    audio = G.synth_audio(c["seconds"], c["seed"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = A.align(mine, audio, g["text"], language="en", regroup=False, suppress_silence=False)
    words = got.all_words()
    assert [(w.word, w.start, w.end, list(w.tokens)) for w in words] == [(w["word"], w["start"], w["end"], w["tokens"]) for w in g["words"]]
    tok = get_tokenizer(False, num_languages=mine.num_languages)
    func = A.make_alignment_func(mine, tok)
    out = func(audio[:480000], [A.WordToken(tok.decode([i]), [i]) for i in g["ids"][:c["b2_words"]]])
    assert [(w["word"], w["start"], w["end"]) for w in out] == [(w["word"], w["start"], w["end"]) for w in g["b2"]]
