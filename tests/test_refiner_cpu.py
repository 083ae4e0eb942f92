"""``refine()``'s bisection (stable_ts_amd/refiner.py) against the reference's ``Refiner``
(stable_whisper/non_whisper/refinement.py), both driven by the SAME synthetic inference function (2-D and 3-D outputs).

* golden: tests/golden/refiner_cases.json.gz = the reference's refined timestamps on 30 seeded cases (1120 inference
  calls); every word start/end and the number of inference calls must be identical.
* live: where /root/reference is importable, more seeds are compared live, including every probe the two
  implementations send to the inference function (segment length, token count, number of un-muted samples).
"""
import gzip
import json
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import make_refiner_golden as mg  # noqa: E402

from stable_ts_amd.refiner import Refiner  # noqa: E402
from stable_ts_amd.result import WhisperResult  # noqa: E402


def test_refiner_matches_reference_golden():
    with gzip.open(os.path.join(HERE, "golden", "refiner_cases.json.gz"), "rb") as f:
        cases = json.loads(f.read().decode("utf-8"))
    assert len(cases) == 30
    moved = 0
    for seed, want in cases.items():
        got, calls = mg.run(Refiner, WhisperResult, int(seed))
        assert got == want["out"], (seed, mg.synth_case(int(seed))[2])
        assert len(calls) == want["n_calls"], seed
        before = [[[w["word"], w["start"], w["end"]] for w in s["words"]] for s in mg.synth_case(int(seed))[1]["segments"]]
        moved += got != before
    assert moved >= 25          # the refinement actually moved timestamps in (nearly) every case


def test_refiner_argument_errors():
    f = mg.make_inference(0, False)
    with pytest.raises(ValueError):
        Refiner(f, steps="sx")
    with pytest.raises(TypeError):
        Refiner(f, bogus=1)
    res = WhisperResult(dict(segments=[dict(start=0.0, end=1.0, text=" a")]))
    with pytest.raises(RuntimeError):
        Refiner(f).refine(torch.ones(16000), res)                        # no word timestamps
    res = WhisperResult([[dict(word=" a", start=0.1, end=0.6, probability=0.9)]])
    with pytest.raises(RuntimeError):
        Refiner(f).refine(torch.ones(16000), res)                        # no tokens and no encode()
    out = Refiner(f).refine(torch.ones(16000), res, encode=lambda s: [7], inplace=False)
    assert out is not res and out.all_words()[0].tokens == [7]
    with pytest.raises(RuntimeError):
        Refiner(lambda a, t: torch.zeros(3, len(t))).refine(torch.ones(16000), out)


@pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")
def test_refiner_matches_reference_live():
    from make_golden import import_reference
    sw = import_reference()
    from stable_whisper.non_whisper.refinement import Refiner as RefRefiner
    for seed in range(500, 520):
        want, ref_calls = mg.run(RefRefiner, sw.WhisperResult, seed, extra=dict(verbose=None))
        got, calls = mg.run(Refiner, WhisperResult, seed)
        assert calls == ref_calls, (seed, mg.synth_case(seed)[2])
        assert got == want, (seed, mg.synth_case(seed)[2])
