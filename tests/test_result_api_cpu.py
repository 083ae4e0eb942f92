"""The rest of the result model's public surface (stable_ts_amd/result.py) next to the reference's own classes
(stable_whisper/result.py): index pickers, ``Segment.add / split / apply_min_dur / add_words``, lock groups, copies,
``WhisperResult.apply_min_dur / adjust_by_silence / adjust_by_result / find`` and the display helpers.  The same seeded
synthetic results (tests/golden/make_regroup_golden.py::synth_result) are built on both sides and every return value
and the resulting state are compared exactly.  Needs /root/reference (live differential test)."""
import copy
import io
import contextlib
import os
import random
import sys
import warnings

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/stable_whisper"), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    import make_golden as G
    G.import_reference()
    import stable_whisper.result as RR
    return RR


def _pair(ref, seed, lock_some=True):
    import make_regroup_golden as mg
    import stable_ts_amd.result as MR
    d = mg.synth_result(seed)
    a, b = ref.WhisperResult(copy.deepcopy(d)), MR.WhisperResult(copy.deepcopy(d))
    if lock_some:
        rnd = random.Random(seed)
        for res in (a, b):
            r2 = random.Random(rnd.random())
            for w in res.all_words():
                x = r2.random()
                if x < 0.08:
                    w.lock_right()
                elif x < 0.14:
                    w.lock_left()
            rnd = random.Random(seed)          # same draws for both
    return a, b


def _state(res):
    return [(s.start, s.end, s.text, s.id, None if not s.has_words else
             [(w.word, w.start, w.end, None if w.probability is None else round(w.probability, 12), w.tokens, w.left_locked, w.right_locked, w.id)
              for w in s.words]) for s in res.segments]


def _try(f):
    """value or the exception type: the reference itself raises on some inputs (e.g. even splits next to a locked last
    boundary) and this package is expected to raise the same way"""
    try:
        return ("ok", f())
    except Exception as e:                                        # noqa: BLE001
        return ("raised", type(e).__name__)


def _seg_state(s):
    return (s.start, s.end, s.text, list(s.tokens), s.temperature, s.avg_logprob, s.compression_ratio, s.no_speech_prob,
            None if not s.has_words else [(w.word, w.start, w.end, w.id, w.left_locked, w.right_locked) for w in s.words])


@pytest.mark.parametrize("seed", range(12))
def test_pickers_and_lock_groups(ref, seed):
    a, b = _pair(ref, seed)
    assert _state(a) == _state(b)
    for sa, sb in zip(a.segments, b.segments):
        assert sa.get_locked_indices() == sb.get_locked_indices()
        assert sa.get_gaps() == sb.get_gaps()
        for g in (0.0, 0.05, 0.3, None):
            assert sa.get_gap_indices(g) == sb.get_gap_indices(g)
        for p in (".", [",", "."], [(",", " "), "?"], [(".", " t")]):
            assert sa.get_punctuation_indices(p) == sb.get_punctuation_indices(p)
        for kw in (dict(max_chars=20), dict(max_words=3), dict(max_chars=25, max_words=4), dict(max_chars=18, even_split=False),
                   dict(max_words=2, include_lock=True), dict(max_chars=15, include_lock=True, ignore_special_periods=True)):
            assert _try(lambda: [int(i) for i in sa.get_length_indices(**kw)]) == _try(lambda: sb.get_length_indices(**kw)), kw
        for kw in (dict(max_dur=1.0), dict(max_dur=0.7, even_split=False), dict(max_dur=0.5, include_lock=True),
                   dict(max_dur=100.0)):
            assert _try(lambda: [int(i) for i in sa.get_duration_indices(**kw)]) == _try(lambda: sb.get_duration_indices(**kw)), kw
        assert sa.words_by_lock() == sb.words_by_lock()
        assert sa.words_by_lock(include_single=True) == sb.words_by_lock(include_single=True)
        assert sa.to_display_str() == sb.to_display_str() and sa.to_display_str(True) == sb.to_display_str(True)
        assert sa.word_count() == sb.word_count() and sa.char_count() == sb.char_count()
    assert a.get_locked_indices() == b.get_locked_indices()
    assert a.get_gaps() == b.get_gaps()
    for g in (0.0, 0.2, 1.0, None):
        assert a.get_gap_indices(g) == b.get_gap_indices(g)
    for p in (".", [",", "?"], [(".", " ")]):
        assert a.get_punctuation_indices(p) == b.get_punctuation_indices(p)
    assert a.all_words_by_lock() == b.all_words_by_lock()
    assert a.all_words_by_lock(by_segment=True, include_single=True) == b.all_words_by_lock(by_segment=True, include_single=True)
    assert [[w.word for w in g] for g in a.all_words_by_lock(only_text=False)] == \
        [[w.word for w in g] for g in b.all_words_by_lock(only_text=False)]
    assert a.segments_to_dicts() == b.segments_to_dicts()
    assert a.to_dict() == b.to_dict()


@pytest.mark.parametrize("seed", range(12))
def test_segment_surgery(ref, seed):
    a, b = _pair(ref, seed)
    rnd = random.Random(100 + seed)
    for sa, sb in zip(a.segments, b.segments):
        n = len(sa.words)
        if n >= 2:
            i = rnd.randrange(n - 1)
            wa, wb = sa.add_words(i, i + 1), sb.add_words(i, i + 1)
            assert (wa.word, wa.start, wa.end, wa.probability, wa.tokens, wa.left_locked, wa.right_locked) == \
                (wb.word, wb.start, wb.end, wb.probability, wb.tokens, wb.left_locked, wb.right_locked)
        cuts = sorted(rnd.sample(range(n), k=min(n, rnd.randrange(0, 4))))
        pa, pb = sa.split(list(cuts)), sb.split(list(cuts))
        assert [_seg_state(x) for x in pa] == [_seg_state(x) for x in pb]
        for md in (0.05, 0.2, 0.5):
            assert _seg_state(sa.apply_min_dur(md)) == _seg_state(sb.apply_min_dur(md))
        ca, cb = copy.deepcopy(sa), copy.deepcopy(sb)
        assert _seg_state(ca) == _seg_state(cb) and cb.words[0] is not sb.words[0] and cb.words[0].tokens is not sb.words[0].tokens
        sh_a, sh_b = copy.copy(sa), copy.copy(sb)
        assert _seg_state(sh_a) == _seg_state(sh_b) and sh_b.words is sb.words
    for k in range(len(a.segments) - 1):
        for kw in (dict(), dict(newline=True), dict(copy_words=True)):
            assert _seg_state(a.segments[k].add(a.segments[k + 1], **kw)) == _seg_state(b.segments[k].add(b.segments[k + 1], **kw))
        sa, sb = a.segments[k] + a.segments[k + 1], b.segments[k] + b.segments[k + 1]
        assert _seg_state(sa) == _seg_state(sb) and sb.words[0] is not b.segments[k].words[0]
    # word-less segments: text / tokens / bounds are concatenated, mixing raises
    d = dict(segments=[dict(start=0.0, end=1.0, text=" a", tokens=[1], avg_logprob=-0.5), dict(start=1.2, end=2.0, text=" b", tokens=[2], avg_logprob=-0.1)])
    import stable_ts_amd.result as MR
    ra, rb = ref.WhisperResult(copy.deepcopy(d)), MR.WhisperResult(copy.deepcopy(d))
    assert _seg_state(ra.segments[0].add(ra.segments[1], newline=True)) == _seg_state(rb.segments[0].add(rb.segments[1], newline=True))
    with pytest.raises(ValueError):
        rb.segments[0].add(b.segments[0])
    with pytest.raises(ValueError):
        rb.segments[0][0]
    # in-place word deletion renumbers
    del a.segments[0][0], b.segments[0][0]
    assert _state(a) == _state(b)


@pytest.mark.parametrize("seed", range(12))
def test_result_min_dur_and_adjust(ref, seed):
    a, b = _pair(ref, seed)
    for md in (0.15, 0.6, 2.5):
        ra, rb = a.apply_min_dur(md), b.apply_min_dur(md)
        assert _state(ra) == _state(rb)
    assert _state(a) == _state(b)                                   # not in place
    a.apply_min_dur(0.3, inplace=True), b.apply_min_dur(0.3, inplace=True)
    assert _state(a) == _state(b)
    # adjust_by_result: the other result has jittered timestamps
    a, b = _pair(ref, seed, lock_some=False)
    oa, ob = _pair(ref, seed, lock_some=False)
    rnd = random.Random(seed)
    for wa, wb in zip(oa.all_words(), ob.all_words()):
        ds, de = rnd.uniform(-0.1, 0.2), rnd.uniform(-0.2, 0.1)
        wa.start, wb.start = wa.start + ds, wb.start + ds
        wa.end, wb.end = max(wa.start, wa.end + de), max(wb.start, wb.end + de)
    out_a, out_b = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(out_a):
        a.adjust_by_result(oa, min_word_dur=0.05, verbose=True)
    with contextlib.redirect_stdout(out_b):
        b.adjust_by_result(ob, min_word_dur=0.05, verbose=True)
    assert _state(a) == _state(b) and out_a.getvalue() == out_b.getvalue()
    ob.all_words()[0].word += "x"
    with pytest.raises(AssertionError):
        b.adjust_by_result(ob)
    # adjust_by_silence on a waveform with silent stretches where the words are
    a, b = _pair(ref, seed, lock_some=False)
    total = int((a.segments[-1].end + 1.0) * 16000)
    g = torch.Generator().manual_seed(seed)
    wav = 0.2 * torch.randn(total, generator=g)
    for _ in range(6):
        s0 = int(torch.randint(0, total - 16000, (1,), generator=g))
        wav[s0: s0 + int(torch.randint(3000, 14000, (1,), generator=g))] = 0
    for kw in (dict(), dict(min_silence_dur=0.3, word_level=False, nonspeech_error=0.1), dict(q_levels=10, k_size=3, use_word_position=False, min_word_dur=0.05)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a.adjust_by_silence(wav.clone(), verbose=None, **kw)
            b.adjust_by_silence(wav.clone(), **kw)
        assert _state(a) == _state(b)
        assert a.nonspeech_sections == b.nonspeech_sections and len(b.nonspeech_sections) > 0
    assert b.adjust_by_silence(torch.zeros(100)) is b              # too short for a mask: untouched
    with pytest.raises(NotImplementedError):
        b.adjust_by_silence(wav, vad=True)


def _matches(m):
    return [(x.text, x.text_match, x.start, x.end, [s.id for s in x.segments], x.word_indices, len(x)) for x in m.matches], m.segment_indices


@pytest.mark.parametrize("seed", range(8))
def test_find(ref, seed):
    a, b = _pair(ref, seed, lock_some=False)
    for pat, kw in ((r" the", {}), (r"[a-z]+\.", {}), (r"\. [A-Za-z]", {}), (r"o\w* \w+", dict(word_level=False)), (r"THE", dict(flags=2)),
                    (r"zzzz", {}), (r"[,.] ", dict(word_level=False))):
        ma, mb = a.find(pat, **kw), b.find(pat, **kw)
        assert _matches(ma) == _matches(mb), pat
        assert bool(ma) == bool(mb) and len(ma) == len(mb)
        if len(mb):
            assert _matches(ma.find(r"\w+", **{k: v for k, v in kw.items() if k != "flags"})) == \
                _matches(mb.find(r"\w+", **{k: v for k, v in kw.items() if k != "flags"}))
            assert mb[0].text == ma[0].text and str(mb[0]).startswith("{")
    # segment-level result: word-level search falls back with a warning
    a.segments[0].convert_to_segment_level(), b.segments[0].convert_to_segment_level()
    with pytest.warns(UserWarning):
        mb = b.find(r"\w+")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert _matches(a.find(r"\w+")) == _matches(mb)


def test_display_rounding_and_deprecations(ref):
    import stable_ts_amd.result as MR
    for cls in (ref, MR):
        w = cls.WordTiming(" hi", 3661.2341, 3662.0004, probability=0.5)
        assert w.to_display_str() == '[01:01:01.234] -> [01:01:02.000] " hi"' and (w.start, w.end, w.duration) == (3661.234, 3662.0, 0.766)
        raw = cls.WordTiming(" hi", 1.23456, 2.34567, round_ts=False)
        assert (raw.start, raw.end) == (1.23456, 2.34567) and raw.round(0.12345) == 0.12345
        with pytest.warns(UserWarning):
            raw.round_all_timestamps()
        assert raw.round(0.12345) == 0.123
        s = cls.Segment(words=[dict(word=" a", start=0.0, end=0.5, probability=1.0, tokens=[1])], ignore_unused_args=True)
        with pytest.warns(UserWarning):
            s.update_seg_with_words()
        with pytest.warns(UserWarning):
            assert s.get_result() is None
        with pytest.warns(UserWarning):
            assert s.words[0].get_segment() is s
        c = copy.copy(w)
        assert (c.word, c.start, c.end, c.probability) == (w.word, w.start, w.end, w.probability) and c.segment is None
    r = MR.WhisperResult(dict(segments=[dict(start=0.0, end=1.0, text=" a")]))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        r.show_regroup_history()
    assert buf.getvalue() == "Result has no history.\n"
    r2 = copy.deepcopy(r)
    assert r2.segments[0] is not r.segments[0] and r2.text == r.text


def test_round3_is_numpy_scalar_rounding_bit_for_bit():
    """stable_ts_amd._num.round3 (the millisecond rounding of every stored timestamp) against the built-in round(x, 3) the
    reference calls: for numpy.float64 inputs that is numpy's rint(x * 1000) / 1000, for Python floats the correctly rounded
    decimal rounding -- same value, same type, same sign of zero, over random values, exact and one-ulp-off ties, already
    rounded values, zeros, negatives, huge and non-finite values."""
    import numpy as np
    from stable_ts_amd._num import round3
    rng = np.random.default_rng(123)
    ties = (rng.integers(0, 4_000_000, size=60000) + 0.5) / 1000.0
    vals = np.concatenate([rng.uniform(0, 4000, size=120000), ties, np.nextafter(ties, 1e9), np.nextafter(ties, -1e9),
                           rng.integers(0, 4_000_000, size=60000) / 1000.0, rng.uniform(-100, 100, size=20000),
                           np.array([0.0, -0.0, 1e-9, -1e-9, 1e15 + 0.3, 123456789.0005, 2.0 ** 52 + 0.5, 1e300, 1e12, 9.99e11,
                                     np.inf, -np.inf])])
    for v in vals:
        x = np.float64(v)
        a, b = round(x, 3), round3(x)
        assert type(a) is type(b) and a.tobytes() == b.tobytes(), (repr(v), repr(a), repr(b))
    assert np.isnan(round3(np.float64("nan")))
    for v in (0.0, 1.0005, 2.675, 1234.5675, -3.14159, 7):
        a, b = round(v, 3), round3(v)
        assert type(a) is type(b) and a == b
    f32 = np.float32(1.23456)
    assert type(round3(f32)) is type(round(f32, 3)) and round3(f32) == round(f32, 3)
