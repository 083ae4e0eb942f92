"""``locate()`` (stable_ts_amd/locator.py) against the reference's ``locate`` (stable_whisper/alignment.py:756-1116) on
the CPU oracle through the engine stand-in (tests/oracle_engine.py): end-time approximation from the alignment matrix,
the greedy duration-window decode with the search text forced in (probability / arg-max / string match, EOT budget,
token budget), word timestamps of the match, and the seek logic between chunks.  Needs /root/reference."""
import contextlib
import io
import os
import sys
import warnings

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")


@pytest.fixture(scope="module")
def models():
    import make_golden as G
    sw = G.import_reference()
    from oracle.whisper.model import build_model
    m = build_model("tiny.en", seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5)
    sw.modify_model(m)
    from oracle_engine import CpuWhisper
    return G, sw, m, CpuWhisper(m)


def _norm(matches):
    out = []
    for x in matches:
        if isinstance(x, dict):
            out.append({k: ([dict(w, probability=round(w["probability"], 7)) for w in v] if k == "duration_window_word" else
                            (round(v, 6) if isinstance(v, float) else v)) for k, v in x.items()})
        else:
            out.append(("segment", x.seek, [(w.word, w.start, w.end, round(float(w.probability), 7), list(w.tokens)) for w in x.words]))
    return out


CASES = [
    dict(text=" aaat", mode=2, count=0),
    dict(text=" aaat aabc", mode=2, count=3, start=2.0, end=70.0),
    dict(text=" aaat", mode=1, count=2, probability_threshold=0.0),
    dict(text=" aaat aabc", mode=1, count=0, probability_threshold=0.0, eots=2, max_token_per_seg=6, duration_window=(2.0, 4.0)),
    dict(text=" aaat", mode=0, count=2, probability_threshold=0.0),
    dict(text=[25, 31], mode=0, count=1, probability_threshold=0.0, exact_token=True, initial_prompt="aabf aabi"),
    dict(text=" aaat", mode=1, count=0, probability_threshold=0.9),                       # never confirmed: seek by chunk
    dict(text=" AAAT", mode=1, count=1, probability_threshold=0.0, case_sensitive=True, suppress_tokens="1,2"),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_locate_matches_reference(models, monkeypatch, case):
    G, sw, ref_model, mine = models
    import stable_ts_amd.locator as L
    from oracle_engine import install
    install(monkeypatch)
    kw = dict(CASES[case])
    text = kw.pop("text")
    audio = G.synth_audio(75.0, seed=11 + case)
    outs = []
    for fn, model, extra in ((sw.alignment.locate, ref_model, dict(verbose=None)), (L.locate, mine, dict(verbose=None))):
        with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
            warnings.simplefilter("ignore")
            try:
                outs.append(_norm(fn(model, audio, text, "en", **kw, **extra)))
            except Exception as e:
                outs.append(("error", type(e).__name__))
    assert outs[0] == outs[1], (CASES[case], outs)


@pytest.mark.parametrize("n", [9000, 52000, 480000])
def test_refinement_callable_matches_reference_seam_b3(models, n):
    """make_refinement_func (seam B3) on the CPU stand-in vs the reference's get_whisper_refinement_func on the same
    oracle model: un-padded spectrogram with the floor from the batch max, pad_or_trim of the mel, one teacher-forced
    pass, softmax over the text vocabulary."""
    G, sw, ref_model, mine = models
    from stable_whisper.alignment import get_whisper_refinement_func
    from stable_ts_amd.alignment import make_refinement_func
    from stable_ts_amd.tokenizer import get_tokenizer
    tok = get_tokenizer(False, num_languages=ref_model.num_languages)
    a = torch.as_tensor(G.synth_audio(31.0, seed=5))[:n]
    seg = torch.stack([a, torch.cat([a[: n // 3], 0.01 * a[n // 3:]])])      # second copy mostly muted: floors differ per item
    ids = tok.encode(" aaat aabc aaau")
    want = get_whisper_refinement_func(ref_model, tok, None, False)(seg, ids)
    got = make_refinement_func(mine, tok)(seg, ids)
    assert tuple(got.shape) == tuple(want.shape) == (2, len(ids), tok.eot)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-7), (got - want).abs().max()
