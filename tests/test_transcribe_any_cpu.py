"""``transcribe_any`` (stable_ts_amd/non_whisper.py, seam B4) next to the reference's
(stable_whisper/non_whisper/transcribe.py) for the branches that run without ffmpeg / torchaudio -- waveform inputs at
the model's rate, every result format, silence suppression, regrouping, ordering options, AudioLoader input, argument
errors -- plus the conversion branches (temporary WAVE file, file bytes, resampling) on this side alone."""
import copy
import os
import sys
import warnings
import wave

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from stable_ts_amd.audio_io import AudioLoader, read_wav  # noqa: E402
from stable_ts_amd.non_whisper import transcribe_any  # noqa: E402
from stable_ts_amd.result import WhisperResult  # noqa: E402

HAVE_REF = os.path.isdir("/root/reference/stable_whisper")


def _state(res):
    return [(s.start, s.end, s.text, None if not s.has_words else [(w.word, w.start, w.end) for w in s.words]) for s in res.segments]


def _wave(seconds, seed):
    g = torch.Generator().manual_seed(seed)
    n = int(seconds * 16000)
    x = 0.2 * torch.randn(n, generator=g)
    for _ in range(8):
        a = int(torch.randint(0, n - 16000, (1,), generator=g))
        x[a: a + int(torch.randint(3000, 12000, (1,), generator=g))] = 0
    return x


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present")
@pytest.mark.parametrize("seed", range(10))
def test_transcribe_any_matches_reference(seed):
    import make_golden as G
    import make_regroup_golden as mg
    G.import_reference()
    from stable_whisper.non_whisper.transcribe import transcribe_any as ref_any
    import stable_whisper.result as RR
    d = mg.synth_result(seed)
    end = max(s["end"] for s in d["segments"]) + 1.0
    wav = _wave(end, seed)
    formats = {
        "dict": lambda: copy.deepcopy(d),
        "segments": lambda: copy.deepcopy(d["segments"]),
        "word_lists": lambda: [copy.deepcopy(s["words"]) for s in d["segments"]],
    }
    seen = {}
    for name, make in formats.items():
        for kw in (dict(), dict(regroup=False), dict(suppress_silence=False, regroup="sg=.3_sl=30"),
                   dict(suppress_word_ts=False, q_levels=10, k_size=3, min_word_dur=0.05, nonspeech_error=0.3),
                   dict(min_silence_dur=0.3, use_word_position=False, force_order=True)):
            outs = []
            for fn, audio in ((ref_any, wav.clone()), (transcribe_any, wav.clone())):
                def inference(audio, tag=None, _fn=fn):
                    seen[_fn] = (type(audio), tag)
                    return make()
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    outs.append(fn(inference, audio, input_sr=16000, inference_kwargs=dict(tag=name), verbose=None, **kw))
            assert _state(outs[0]) == _state(outs[1]), (name, kw)
            assert outs[0].regroup_history == outs[1].regroup_history
            assert outs[0].nonspeech_sections == outs[1].nonspeech_sections
            assert outs[0].to_dict() == outs[1].to_dict()
            assert seen[ref_any] == seen[transcribe_any] == (torch.Tensor, name)
    # numpy in -> numpy out to the function; a result object is passed through; AudioLoader input warns and skips silence
    for fn, R, L in ((ref_any, RR.WhisperResult, None), (transcribe_any, WhisperResult, None)):
        got = {}

        def inference(audio):
            got["type"] = type(audio)
            return R(copy.deepcopy(d))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = fn(inference, wav.numpy().copy(), input_sr=16000, model_sr=16000, audio_type="numpy", verbose=None)
        assert got["type"] is np.ndarray and isinstance(r, R)
        seen[fn] = _state(r)
    assert seen[ref_any] == seen[transcribe_any]


def test_transcribe_any_conversions_and_errors(tmp_path):
    d = dict(segments=[dict(start=0.2, end=1.4, text=" hello world", words=[
        dict(word=" hello", start=0.2, end=0.8, probability=0.9, tokens=[1]),
        dict(word=" world", start=0.8, end=1.4, probability=0.8, tokens=[2])])])
    wav = _wave(2.0, 3)
    got = {}

    def inference(audio, **kw):
        got["audio"] = audio
        return copy.deepcopy(d)

    tmp = str(tmp_path / "t.wav")
    r = transcribe_any(inference, wav, input_sr=16000, audio_type="str", temp_file=tmp, suppress_silence=False)
    assert got["audio"] == os.path.abspath(tmp) and not os.path.exists(tmp) and isinstance(r, WhisperResult)   # written, used, removed
    r = transcribe_any(inference, wav, input_sr=16000, audio_type="byte", suppress_silence=False)
    data, sr = read_wav(got["audio"])
    assert sr == 16000 and np.abs(data[:, 0] - wav.numpy()).max() <= 0.5 / 32768 + 1e-7
    r = transcribe_any(inference, wav, input_sr=16000, model_sr=8000, audio_type="numpy")
    assert isinstance(got["audio"], np.ndarray) and got["audio"].shape == (16000,)          # resampled for the model
    assert len(r.nonspeech_sections) > 0                                                    # silence from the 16 kHz input
    # file input: handed over as a path; bytes of it as a temp file when the function wants a path
    p = str(tmp_path / "in.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1), w.setsampwidth(2), w.setframerate(16000)
        w.writeframes((wav.numpy() * 32767).astype("<i2").tobytes())
    r = transcribe_any(inference, p)
    assert got["audio"] == p and len(r.nonspeech_sections) > 0
    r = transcribe_any(inference, p, audio_type="torch", model_sr=16000)
    assert torch.is_tensor(got["audio"]) and got["audio"].shape == wav.shape
    r = transcribe_any(inference, open(p, "rb").read(), audio_type="str", temp_file=tmp, suppress_silence=False)
    assert got["audio"] == os.path.abspath(tmp) and not os.path.exists(tmp)
    r = transcribe_any(inference, p, only_voice_freq=True, audio_type="numpy", model_sr=16000)
    assert isinstance(got["audio"], np.ndarray)
    with pytest.warns(UserWarning):
        r = transcribe_any(inference, AudioLoader(wav), only_voice_freq=True)
    assert isinstance(got["audio"], AudioLoader) and r.nonspeech_sections == []
    # the function's exception propagates and the temp file is still removed
    def boom(audio):
        raise KeyError("x")
    with pytest.raises(KeyError):
        transcribe_any(boom, wav, input_sr=16000, audio_type="str", temp_file=tmp)
    assert not os.path.exists(tmp)
    for bad, exc in ((dict(audio_type="mp3"), NotImplementedError), (dict(), ValueError), (dict(input_sr=16000, denoiser="demucs"), NotImplementedError),
                     (dict(input_sr=16000, vad=True), NotImplementedError)):
        with pytest.raises(exc):
            transcribe_any(inference, wav, **bad)
    with pytest.raises(ValueError):
        transcribe_any(inference, p, audio_type="numpy")
    with pytest.raises(ValueError):
        transcribe_any(inference, AudioLoader(wav), audio_type="torch")
    with pytest.raises(TypeError):
        transcribe_any(inference, [1, 2, 3])
