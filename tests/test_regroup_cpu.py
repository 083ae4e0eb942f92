"""Regrouping layer (stable_ts_amd/result.py + regroup.py) against the reference's own ``WhisperResult``.

* golden: tests/golden/regroup_cases.json.gz holds outputs of /root/reference's ``WhisperResult.regroup`` on seeded
  synthetic results (tests/golden/make_regroup_golden.py); every word, timestamp, lock flag, segment statistic and the
  regroup history must be identical (these are exact list operations on millisecond-rounded floats: no tolerance).
* live: where /root/reference is importable (this container) a wider randomized differential test runs as well.
"""
import contextlib
import copy
import gzip
import io
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from make_regroup_golden import ALGOS, snapshot, synth_result  # noqa: E402

from stable_ts_amd.result import Segment, WhisperResult, WordTiming  # noqa: E402


def _golden():
    with gzip.open(os.path.join(HERE, "golden", "regroup_cases.json.gz"), "rb") as f:
        return json.loads(f.read().decode("utf-8"))


def _run(inp, algo):
    res = WhisperResult(copy.deepcopy(inp))
    with contextlib.redirect_stdout(io.StringIO()):
        res.regroup(algo)
    return json.loads(json.dumps(snapshot(res)))


def test_regroup_matches_reference_golden():
    g = _golden()
    assert len(g["cases"]) >= 140
    for seed, inp in g["inputs"].items():
        assert synth_result(int(seed)) == inp        # the generator is deterministic on this interpreter
    for c in g["cases"]:
        if "error" in c:                               # the reference itself fails on this input: same failure expected
            with pytest.raises(Exception) as ei:
                _run(g["inputs"][str(c["seed"])], c["algo"])
            assert type(ei.value).__name__ == c["error"], (c["seed"], c["algo"])
            continue
        got = _run(g["inputs"][str(c["seed"])], c["algo"])
        assert got["history"] == c["out"]["history"], (c["seed"], c["algo"])
        assert got["text"] == c["out"]["text"], (c["seed"], c["algo"])
        assert len(got["segments"]) == len(c["out"]["segments"]), (c["seed"], c["algo"])
        for a, b in zip(got["segments"], c["out"]["segments"]):
            assert a == b, (c["seed"], c["algo"], a, b)


def test_default_regroup_program():
    # "da" expands to isp_cm_sp=.* /。/?/？_sg=.5_sp=,* /，++++50_sl=70_cm (result.py:3008)
    words = []
    t = 0.0
    for i, w in enumerate(" Hello there, Mr. Smith. How are you today? I am fine".split(" ")[1:]):
        words.append(dict(word=" " + w, start=t, end=t + 0.3, probability=0.5, tokens=[i]))
        t += 0.3 + (0.8 if w == "today?" else 0.0)
    res = WhisperResult(dict(language="en", segments=[dict(start=0, end=t, text="", seek=0.0, words=words)]))
    res.regroup(True)
    # "Mr." is a special period (no split); ". " after Smith and "? " split; the comma split needs >= 50 chars
    assert [s.text for s in res] == [" Hello there, Mr. Smith.", " How are you today?", " I am fine"]
    assert res.regroup_history == "isp=1_cm=2.5+++0_sp=.* /。/?/？+0+0++++1_sg=0.5+0+0+1_sp=,* /，+0+0++50++1" \
                                  "_sl=70++1+0+0+0+0+1_cm=2.5+++0"
    assert [w.segment_id for w in res.all_words()] == [0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2]
    assert res.regroup(False) is res
    with pytest.raises(NotImplementedError):
        res.regroup("xyz=1")


def test_result_schema_and_rounding():
    w = WordTiming(" a", 0.12345, 0.98765, probability=0.5, tokens=[3])
    assert (w.start, w.end, w.duration, len(w)) == (0.123, 0.988, 0.865, 2)
    w.clamp_max(0.5, clip_start=True)
    assert w.start == 0.488
    seg = Segment(start=5.0, end=6.0, text="ignored", words=[w.to_dict(), dict(word=" b", start=1.0, end=1.5, tokens=[4])])
    assert seg.start == 0.488 and seg.end == 1.5 and seg.text == " a b" and seg.tokens == [3, 4]
    d = seg.to_dict()
    assert list(d) == ["start", "end", "text", "seek", "tokens", "temperature", "avg_logprob", "compression_ratio",
                       "no_speech_prob", "words"]
    assert list(d["words"][0]) == ["word", "start", "end", "probability", "tokens"]            # result.py:192-204
    res = WhisperResult([seg.to_dict()])
    assert list(res.to_dict()) == ["text", "segments", "language", "ori_dict", "regroup_history", "nonspeech_sections",
                                   "unfinished"]
    # list-of-word-lists input (result.py:977-989) and segment-level conversion
    res2 = WhisperResult([[dict(word=" x", start=0.0, end=0.2)], [dict(word=" y", start=0.3, end=0.4)]])
    assert [s.text for s in res2] == [" x", " y"]
    res2.merge_all_segments().convert_to_segment_level()
    assert res2[0].words is None and res2[0].text == " x y" and (res2[0].start, res2[0].end) == (0.0, 0.4)
    assert res2.regroup_history == "ms_csl"


@pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")
def test_regroup_matches_reference_live():
    from make_golden import import_reference
    sw = import_reference()
    import random
    rng = random.Random(1234)
    n = n_err = 0
    for seed in range(100, 160):
        inp = synth_result(seed)
        for algo in rng.sample(ALGOS, 4):
            ref = sw.WhisperResult(copy.deepcopy(inp))
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    ref.regroup(algo)
            except Exception as e:                  # e.g. even-splitting a one-word segment: same failure expected
                with pytest.raises(type(e)):
                    _run(inp, algo)
                n_err += 1
                continue
            want = json.loads(json.dumps(snapshot(ref)))
            got = _run(inp, algo)
            assert got == want, (seed, algo)
            n += 1
    assert n + n_err == 240 and n >= 200


@pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")
def test_fill_in_gaps_and_method_level_edits_live():
    """fill_in_gaps needs a second result object (no DSL string form without a content cache), remove_words_by_str
    accepts a callable filter, custom_operation accepts callables: compared through the methods."""
    from make_golden import import_reference
    sw = import_reference()
    for seed in range(300, 330):
        a, b = synth_result(seed), synth_result(seed + 1000)
        # thin out `a` so that it has gaps the words of `b` can fall into
        a["segments"] = [s for k, s in enumerate(a["segments"]) if k % 2 == 0]
        outs = []
        for cls in (sw.WhisperResult, WhisperResult):
            ra, rb = cls(copy.deepcopy(a)), cls(copy.deepcopy(b))
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    ra.fill_in_gaps(rb, min_gap=0.3, verbose=False)
                    ra.remove_words_by_str([" the", " fox,"], filters=lambda w: w.duration < 0.5, verbose=False)
                    ra.custom_operation("word", lambda x, y: x.strip().startswith(y), "d", "lockright")
                    ra.split_by_length(max_words=5)
                snap = json.loads(json.dumps(snapshot(ra)))
                import re
                snap["history"] = re.sub(r"0x[0-9a-f]+", "ADDR", snap["history"]).replace("stable-whisper", "PKG").replace("stable-ts-amd", "PKG")   # object reprs
                outs.append(snap)
            except Exception as e:
                outs.append(type(e).__name__)
        assert outs[0] == outs[1], seed
