"""CPU-only tests: C-ABI library loads and exports every symbol of include/swx.h, weight generators agree, host logic."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from stable_ts_amd import _lib
    lib = _lib.load(build_if_missing=True)
    hdr = open(os.path.join(ROOT, "include", "swx.h")).read()
    names = set(re.findall(r"\b(swx_[a-z0-9_]+)\s*\(", hdr))
    assert names, "no prototypes found"
    for n in sorted(names):
        assert hasattr(lib, n), f"libswx.so does not export {n}"
        assert n in _lib.SYMBOLS, f"ctypes table lacks {n}"
    assert lib.swx_version() >= 1
    assert lib.swx_strerror(-4).decode().startswith("gemm")


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "stable_ts_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_no_cpu_path():
    import stable_ts_amd as sw
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        sw.load_model("tiny.en", weights="random")
    with pytest.raises(RuntimeError):
        sw.load_model("tiny.en", device="cpu", weights="random")


def test_random_state_dict_matches_oracle_generator():
    import stable_ts_amd as sw
    from oracle.whisper import model as om
    for name in ("tiny.en", "base"):
        a = om.random_state_dict(om.dims_for(name), 7, 0.02, 3.0, 0.5)
        b = sw.random_state_dict(sw.dims_for(name), 7, 0.02, 3.0, 0.5)
        assert list(a) == list(b)
        assert all(torch.equal(a[k], b[k]) for k in a)


def test_mel_filterbank_matches_oracle():
    from oracle.whisper.audio import _mel_filters_np
    from stable_ts_amd.audio import slaney_mel_filterbank
    for n in (80, 128):
        assert np.array_equal(_mel_filters_np(n), slaney_mel_filterbank(n))


def test_tokenizer_roundtrip_and_ids():
    from stable_ts_amd.tokenizer import get_tokenizer
    en = get_tokenizer(False, num_languages=99)
    assert (en.eot, en.sot, en.no_timestamps, en.timestamp_begin) == (50256, 50257, 50362, 50363)
    assert en.timestamp_begin + 1500 == 51863
    ml = get_tokenizer(True, num_languages=100, language="en", task="transcribe")
    assert (ml.eot, ml.sot, ml.no_timestamps, ml.timestamp_begin) == (50257, 50258, 50364, 50365)
    assert ml.sot_sequence == (50258, 50259, 50360)
    ids = [19, 18, 3, 17, 40000, 21, 0, 16, 25]
    assert en.encode(en.decode(ids)) == ids
    assert en.encode(" ...") == [17] and en.encode(" ") == [16]


def test_split_tokens_and_punctuation_merge_match_oracle_glue():
    from oracle import stable as ost
    from stable_ts_amd import timing as pt
    from stable_ts_amd.tokenizer import get_tokenizer
    tok = get_tokenizer(False, num_languages=99)
    rng = np.random.default_rng(0)
    for _ in range(20):
        toks = rng.integers(0, 50000, size=rng.integers(1, 30)).tolist()
        assert pt._split_tokens(list(toks), tok) == ost.split_tokens(list(toks), tok)
    segs = [dict(tokens=[50363, 19, 20, 0, 50400], seek=0.0), dict(tokens=[50400, 40, 21, 1, 50500], seek=0.0)]
    a = pt.split_word_tokens([dict(s) for s in segs], tok, padding=" ...")
    b = ost.split_word_tokens([dict(s) for s in segs], tok, padding=" ...")
    assert a == b


def test_stabilization_matches_reference_semantics():
    from stable_ts_amd.stabilization import NonSpeechPredictor, mask2timing, suppress_segment_silence, wav2mask
    t = torch.arange(16000 * 10) / 16000.0
    x = 0.4 * torch.sin(2 * np.pi * 300 * t)
    x[16000 * 3: 16000 * 5] = 0
    m = wav2mask(x)
    s, e = mask2timing(m)
    assert len(s) == 1 and abs(s[0] - 3.0) < 0.1 and abs(e[0] - 5.0) < 0.1
    p = NonSpeechPredictor(get_mask=True).predict(x, offset=10.0)
    assert not p["is_silent"] and p["mask"].shape == (1501,) and abs(p["timings"][0][0] - 13.0) < 0.1
    assert NonSpeechPredictor().predict(torch.zeros(16000 * 5))["is_silent"]
    seg = dict(start=2.0, end=6.0, words=[dict(word=" a", start=2.0, end=3.5), dict(word=" b.", start=3.5, end=6.0)])
    suppress_segment_silence(seg, s, e, min_word_dur=0.1)
    assert seg["words"][0]["start"] == 2.0 and seg["words"][1]["end"] == 6.0


def _probe_cases():
    g = torch.Generator().manual_seed(11)
    t = torch.arange(560000) / 16000.0
    speech = 0.3 * torch.sin(2 * np.pi * 220 * t) * (torch.sin(2 * np.pi * 0.4 * t) > -0.2) + 0.004 * torch.randn(560000, generator=g)
    return [speech[:480000], speech[80000:], speech, speech[:1234], speech[:999], speech[:12345] * 1e-7,
            torch.randn(333333, generator=g) * 0.1, torch.zeros(480000), speech[:480001], speech[:641]]


def test_loudness_from_probe_is_the_full_length_path_bit_for_bit():
    """The host half of the device silence analysis (stabilization.loudness_from_probe) on probe values produced here by
    numpy (k-th largest |x| by np.partition, |x| at probe_indices): the loudness curve, and everything NonSpeechPredictor
    derives from it, must equal the full-length path's exactly; the GPU test checks the kernel against the same numpy
    expressions (tests/test_gpu_kernels.py::test_loudness_probe_kernel)."""
    from stable_ts_amd.stabilization import NonSpeechPredictor, audio2loudness, loudness_from_probe, probe_indices
    n_equal = 0
    for x in _probe_cases():
        n = x.numel()
        ref = audio2loudness(x.clone())
        idx = probe_indices(n)
        if idx is None:
            assert ref is None
            continue
        assert idx.min() >= 0 and idx.max() < n
        k = int(n * 0.001)
        ax = x.abs().numpy()
        thr = float(np.partition(ax, ax.size - k)[ax.size - k]) if k else float("nan")
        got = loudness_from_probe(n, thr, idx, torch.from_numpy(ax[idx]))
        if not k:
            assert got is False                  # the quantile branch stays on the full path
            continue
        assert torch.equal(got, ref), n
        a, b = NonSpeechPredictor(get_mask=True), NonSpeechPredictor(get_mask=True)
        r1, r2 = a.predict(x.clone(), offset=3.0), b.predict(None, offset=3.0, loud=got)
        assert (r1["timings"] is None) == (r2["timings"] is None)
        assert r1["timings"] is None or np.array_equal(r1["timings"], r2["timings"])
        assert (r1["mask"] is None) == (r2["mask"] is None) and (r1["mask"] is None or torch.equal(r1["mask"], r2["mask"]))
        assert r1["is_silent"] == r2["is_silent"] and a.timings() == b.timings()
        n_equal += 1
    assert n_equal >= 7
    # a sample the interpolation reads but the probe did not deliver must be noticed, not silently read as something else
    x = _probe_cases()[0]
    idx = probe_indices(x.numel())
    short = idx.copy()
    short[4 * 700: 4 * 700 + 4] = short[0]
    import stable_ts_amd.stabilization as st
    st._probe_scratch.buf = {}
    ax = x.abs().numpy()
    assert loudness_from_probe(x.numel(), float(np.partition(ax, ax.size - 480)[ax.size - 480]), short, torch.from_numpy(ax[short])) is False
    st._probe_scratch.buf = {}


def test_loudness_from_probe_random_lengths():
    """the reconstruction for 40 random window lengths (the last window of a recording is ragged; clip sections and
    nonspeech_skip cut windows anywhere): always the full-length curve bit for bit, never a NaN refusal"""
    from stable_ts_amd.stabilization import audio2loudness, loudness_from_probe, probe_indices
    g = torch.Generator().manual_seed(2024)
    lens = [int(v) for v in torch.randint(1000, 480001, (36,), generator=g)] + [1000, 1001, 479999, 480000]
    for n in lens:
        x = torch.randn(n, generator=g) * (10.0 ** float(torch.empty(1).uniform_(-4, 0, generator=g)))
        x[torch.rand(n, generator=g) < 0.3] *= 0.01
        ref = audio2loudness(x.clone())
        idx = probe_indices(n)
        if idx is None:
            assert ref is None
            continue
        ax = x.abs().numpy()
        k = int(n * 0.001)
        thr = float(np.partition(ax, ax.size - k)[ax.size - k])
        got = loudness_from_probe(n, thr, idx, torch.from_numpy(ax[idx]))
        assert got is not False and torch.equal(got, ref), n


def test_entry_points_park_the_intra_op_pool():
    """model.transcribe / align / ... run with torch's intra-op pool parked (stabilization.host_single_thread) and give it
    back afterwards, also when the call raises; a model object that computes on the host (the test stand-in) keeps it"""
    from stable_ts_amd.stabilization import host_single_thread
    before = torch.get_num_threads()
    if before == 1:
        torch.set_num_threads(2)
    n0 = torch.get_num_threads()
    try:
        seen = []

        @host_single_thread
        def entry(model, fail=False):
            seen.append(torch.get_num_threads())
            if fail:
                raise ValueError("x")
            return 7

        class Dev:
            pass

        class Host:
            computes_on_host = True

        assert entry(Dev()) == 7 and seen[-1] == 1 and torch.get_num_threads() == n0
        with pytest.raises(ValueError):
            entry(Dev(), fail=True)
        assert torch.get_num_threads() == n0
        assert entry(Host()) == 7 and seen[-1] == n0
        import stable_ts_amd.alignment as A
        import stable_ts_amd.transcribe as T
        for fn in (T.transcribe_stable, T.transcribe_minimal, A.align, A.align_words, A.refine):
            assert hasattr(fn, "__wrapped__"), fn.__name__
    finally:
        torch.set_num_threads(before)


def test_result_container():
    from stable_ts_amd.result import UnsortedException, WhisperResult
    r = WhisperResult(dict(language="en", segments=[dict(start=0.0, end=1.0, text=" a b", seek=0.0, tokens=[1, 2], words=[
        dict(word=" a", start=0.0, end=0.5, probability=0.9, tokens=[1]), dict(word=" b", start=0.5, end=1.0, probability=0.8, tokens=[2])])]))
    d = r.to_dict()
    assert d["text"] == " a b" and d["segments"][0]["words"][1]["end"] == 1.0 and r.has_words and len(r.all_words()) == 2
    with pytest.raises(UnsortedException):
        WhisperResult(dict(segments=[dict(start=2.0, end=1.0, text="x")]))


def test_model_decoder_and_call_surface():
    """model.decoder(tokens, xa) / model(mel, tokens) (seam B1): argument normalisation in front of the device call"""
    import types
    import pytest
    import torch
    from stable_ts_amd.model import Whisper
    seen = {}

    class Fake:
        def logits(self, toks, xa):
            seen["toks"], seen["xa"] = toks, tuple(xa.shape)
            return torch.zeros(len(toks), len(toks[0]), 7)

        def encoder(self, mel):
            return torch.zeros(mel.shape[0] if mel.ndim == 3 else 1, 1500, 8)

    f = Fake()
    f.decoder = types.MethodType(Whisper.decoder, f)
    out = Whisper.decoder(f, torch.tensor([[1, 2, 3]]), torch.zeros(1, 1500, 8))
    assert out.shape == (1, 3, 7) and seen["toks"] == [[1, 2, 3]]
    Whisper.decoder(f, torch.tensor([4, 5]), torch.zeros(1500, 8))                 # 1-D tokens, 2-D features
    assert seen["toks"] == [[4, 5]] and seen["xa"] == (1, 1500, 8)
    Whisper.decoder(f, [[1, 2], [3, 4]], torch.zeros(1, 1500, 8))                  # one window shared by two rows
    assert seen["xa"] == (2, 1500, 8)
    with pytest.raises(ValueError):
        Whisper.decoder(f, [[1], [2], [3]], torch.zeros(2, 1500, 8))
    with pytest.raises(NotImplementedError):
        Whisper.decoder(f, [[1]], torch.zeros(1, 1500, 8), kv_cache={1: 2})
    assert Whisper.__call__(f, torch.zeros(2, 80, 3000), torch.tensor([[1, 2], [3, 4]])).shape == (2, 2, 7)
    with pytest.raises(NotImplementedError):
        Whisper.install_kv_cache_hooks(f)


def test_ctypes_table_matches_header_prototypes():
    """every prototype of include/swx.h, parameter by parameter, against the ctypes signature the Python host binds
    (stable_ts_amd/_lib.py): count, pointer-ness, integer width, float"""
    import ctypes
    from stable_ts_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "swx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    protos = re.findall(r"\n\s*([A-Za-z_][\w \*]*?)\b(swx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr)
    assert len(protos) >= 25

    def kind(decl: str) -> str:
        decl = decl.strip()
        if "*" in decl:
            return "ptr"
        base = re.sub(r"\b(const|unsigned|struct)\b", "", decl).split()[0]
        return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "size_t": "size", "float": "f32", "uint64_t": "u64"}[base]

    def ckind(t) -> str:
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or getattr(t, "_type_", None) not in (None, "i", "l", "q", "f", "L", "Q", "I"):
            return "ptr"
        return {ctypes.c_int: "i32", ctypes.c_int32: "i32", ctypes.c_int64: "i64", ctypes.c_size_t: "size", ctypes.c_float: "f32",
                ctypes.c_uint64: "u64"}[t]

    checked = 0
    for ret, name, params in protos:
        if name not in _lib.SYMBOLS:
            continue
        res, args = _lib.SYMBOLS[name]
        plist = [p for p in (x.strip() for x in params.split(",")) if p and p != "void"]
        assert len(plist) == len(args), (name, plist, args)
        for decl, t in zip(plist, args):
            k = kind(decl)
            if k == "size":
                assert t is ctypes.c_size_t, (name, decl)
            else:
                assert ckind(t) == k, (name, decl, t)
        if ret.strip() == "void":
            assert res is None, name
            checked += 1
            continue
        rk = "ptr" if "*" in ret else kind(ret)
        if rk == "ptr":
            assert res in (ctypes.c_char_p, ctypes.c_void_p), name
        elif rk == "size":
            assert res is ctypes.c_size_t, name
        elif rk == "i64":
            assert res is ctypes.c_int64, name
        else:
            assert res is ctypes.c_int, name
        checked += 1
    assert checked == len(_lib.SYMBOLS), (checked, len(_lib.SYMBOLS))


def test_header_is_plain_c():
    """include/swx.h is the drop-in boundary: it must parse as C (and C++) on its own, no torch / HIP types"""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, "include", "swx.h")
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    for lang, std in (("c", "-std=c11"), ("c++", "-std=c++17")):
        r = subprocess.run(["gcc", "-fsyntax-only", "-x", lang, std, "-Wall", hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    text = open(hdr).read()
    assert "torch" not in text.lower().replace("pytorch", "") and "hipStream_t" not in text.split("*/")[-1]


def test_word_means_equal_numpy_mean_per_word_bit_for_bit():
    """timing._word_means groups the words of a window by token count and sums column by column instead of calling np.mean per
    word (timing.py:292-295 upstream): same float64 additions in the same order -- equal bits, equal scalar type, NaN (and numpy's
    warning) for an empty word, np.mean itself from 8 tokens on (numpy's pairwise blocking).  20 000 random windows."""
    import warnings

    import numpy as np
    from stable_ts_amd.timing import _word_means
    rng = np.random.default_rng(0)
    for trial in range(20000):
        cnts = rng.choice([0, 1, 1, 1, 2, 2, 3, 4, 5, 6, 7, 8, 9, 15], size=rng.integers(0, 12))
        bounds = np.pad(np.cumsum(cnts), (1, 0)).astype(np.int64)
        pa = rng.random(int(bounds[-1]) + rng.integers(0, 4)) * rng.choice([1.0, 1e-8, 1e-3])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = [np.mean(pa[i:j]) for i, j in zip(bounds[:-1], bounds[1:])]
            got = _word_means(pa, bounds)
        assert len(ref) == len(got)
        for a, b in zip(ref, got):
            assert type(a) is type(b) and (a == b or (np.isnan(a) and np.isnan(b))), (cnts, a, b)

