"""Pin the CPU oracle (oracle/whisper/*) against independent implementations installed in this image.

The reference holds no golden vectors for the hot path, and openai-whisper is
not installable offline, so the oracle is pinned against:
  * transformers.WhisperFeatureExtractor        -> mel filterbank + log-mel
  * transformers WhisperModel (same weights)     -> encoder output, decoder logits, cross-attn probabilities
  * transformers generation_whisper._median_filter / _dynamic_time_warping (verbatim ports of upstream)
  * the known-answer DTW vector of SURVEY.md section 8c
"""
import numpy as np
import pytest
import torch

from oracle.whisper import audio as oa
from oracle.whisper import model as om
from oracle.whisper import timing as ot


def _signal(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 16000.0
    x = 0.3 * torch.sin(2 * np.pi * 440 * t) * (0.5 + 0.5 * torch.sin(2 * np.pi * 3 * t))
    x += 0.1 * torch.sin(2 * np.pi * 1234.5 * t) + 0.01 * torch.randn(n, generator=g)
    return x.float()


@pytest.mark.parametrize("n_mels", [80, 128])
def test_mel_filters_match_hf(n_mels):
    from transformers.audio_utils import mel_filter_bank
    hf = mel_filter_bank(num_frequency_bins=201, num_mel_filters=n_mels, min_frequency=0.0,
                         max_frequency=8000.0, sampling_rate=16000, norm="slaney", mel_scale="slaney")
    ours = oa._mel_filters_np(n_mels)
    assert ours.shape == (n_mels, 201)
    np.testing.assert_allclose(ours, hf.T.astype(np.float32), rtol=0, atol=1e-7)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_log_mel_matches_hf(n_mels):
    from transformers import WhisperFeatureExtractor
    x = _signal(480000)
    fe = WhisperFeatureExtractor(feature_size=n_mels)
    hf = fe(x.numpy(), sampling_rate=16000, return_tensors="np")["input_features"][0]
    ours = oa.log_mel_spectrogram(x, n_mels).numpy()
    assert ours.shape == (n_mels, 3000)
    np.testing.assert_allclose(ours, hf, rtol=0, atol=2e-4)


def test_log_mel_padding_contract():
    # the callers guarantee len + padding == 480000 (original_whisper.py:528-530)
    x = _signal(123457)
    m = oa.log_mel_spectrogram(x, 80, padding=480000 - x.numel())
    assert m.shape == (80, 3000)
    m2 = oa.log_mel_spectrogram(oa.pad_or_trim(x, 480000), 80)
    assert torch.equal(m, m2)


def test_median_filter_matches_hf_port():
    from transformers.models.whisper.generation_whisper import _median_filter
    x = torch.randn(3, 17, 211, generator=torch.Generator().manual_seed(1))
    assert torch.equal(ot.median_filter(x, 7), _median_filter(x[None], 7)[0])


def test_dtw_known_answer():
    ti, tj = ot.dtw_cpu_py(np.zeros((3, 5)))
    assert ti.tolist() == [0, 1, 2, 2, 2, 2, 2]
    assert tj.tolist() == [0, 0, 0, 1, 2, 3, 4]
    ti, tj = ot.dtw_cpu(np.zeros((3, 5)))
    assert ti.tolist() == [0, 1, 2, 2, 2, 2, 2]
    assert tj.tolist() == [0, 0, 0, 1, 2, 3, 4]


@pytest.mark.parametrize("shape,quant", [((7, 31), None), ((23, 57), 4), ((40, 90), 2), ((1, 9), None), ((9, 1), None)])
def test_dtw_matches_hf_port(shape, quant):
    from transformers.models.whisper.generation_whisper import _dynamic_time_warping
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    x = rng.standard_normal(shape).astype(np.float32)
    if quant:  # tie-heavy
        x = np.round(x * quant) / quant
    ref_i, ref_j = _dynamic_time_warping(x.astype(np.float64))
    for fn in (ot.dtw_cpu_py, ot.dtw_cpu):
        ti, tj = fn(x.astype(np.float64))
        assert ti.tolist() == ref_i.tolist() and tj.tolist() == ref_j.tolist()


def _hf_model_from(oracle_model):
    from transformers import WhisperConfig, WhisperModel
    d = oracle_model.dims
    cfg = WhisperConfig(vocab_size=d.n_vocab, num_mel_bins=d.n_mels, d_model=d.n_audio_state,
                        encoder_layers=d.n_audio_layer, encoder_attention_heads=d.n_audio_head,
                        decoder_layers=d.n_text_layer, decoder_attention_heads=d.n_text_head,
                        encoder_ffn_dim=4 * d.n_audio_state, decoder_ffn_dim=4 * d.n_text_state,
                        max_source_positions=d.n_audio_ctx, max_target_positions=d.n_text_ctx,
                        attn_implementation="eager", dropout=0.0, attention_dropout=0.0,
                        activation_dropout=0.0)
    hf = WhisperModel(cfg).eval()
    # name map: reference whisper_word_level/hf_whisper.py:30-51 (vanilla -> HF)
    sub = [("blocks", "layers"), ("mlp.0", "fc1"), ("mlp.2", "fc2"), ("mlp_ln", "final_layer_norm"),
           (".attn.query", ".self_attn.q_proj"), (".attn.key", ".self_attn.k_proj"),
           (".attn.value", ".self_attn.v_proj"), (".attn_ln", ".self_attn_layer_norm"),
           (".attn.out", ".self_attn.out_proj"), (".cross_attn.query", ".encoder_attn.q_proj"),
           (".cross_attn.key", ".encoder_attn.k_proj"), (".cross_attn.value", ".encoder_attn.v_proj"),
           (".cross_attn_ln", ".encoder_attn_layer_norm"), (".cross_attn.out", ".encoder_attn.out_proj"),
           ("decoder.ln.", "decoder.layer_norm."), ("encoder.ln_post.", "encoder.layer_norm."),
           ("token_embedding", "embed_tokens"), ("encoder.positional_embedding", "encoder.embed_positions.weight"),
           ("decoder.positional_embedding", "decoder.embed_positions.weight")]
    new = {}
    for k, v in oracle_model.state_dict().items():
        for a, b in sub:
            k = k.replace(a, b)
        new[k] = v
    missing, unexpected = hf.load_state_dict(new, strict=False)
    assert not unexpected, unexpected
    assert all("k_proj.bias" in m for m in missing) or not missing, missing
    return hf


def test_model_matches_hf():
    dims = om.ModelDimensions(n_mels=80, n_audio_ctx=100, n_audio_state=64, n_audio_head=2, n_audio_layer=2,
                              n_vocab=51864, n_text_ctx=64, n_text_state=64, n_text_head=2, n_text_layer=2)
    m = om.build_model(dims, seed=7, std=0.08)
    hf = _hf_model_from(m)
    g = torch.Generator().manual_seed(3)
    mel = torch.randn(1, 80, 200, generator=g)
    tokens = torch.randint(0, 50000, (1, 11), generator=g)
    with torch.no_grad(), om.disable_sdpa():
        xa = m.encoder(mel)
        qks = []
        hooks = [b.cross_attn.register_forward_hook(lambda _, i, o: qks.append(o[-1])) for b in m.decoder.blocks]
        logits = m.decoder(tokens, xa)
        for h in hooks:
            h.remove()
        out = hf(input_features=mel, decoder_input_ids=tokens, output_attentions=True)
    torch.testing.assert_close(xa, out.encoder_last_hidden_state, rtol=1e-4, atol=1e-5)
    hf_logits = out.last_hidden_state @ hf.decoder.embed_tokens.weight.T
    torch.testing.assert_close(logits, hf_logits, rtol=1e-4, atol=1e-4)
    for qk, att in zip(qks, out.cross_attentions):
        torch.testing.assert_close(qk.softmax(-1), att, rtol=1e-4, atol=1e-6)
    # sdpa path == explicit path
    with torch.no_grad():
        logits2 = m.decoder(tokens, m.encoder(mel))
    torch.testing.assert_close(logits, logits2, rtol=1e-4, atol=1e-5)


def test_hf_checkpoint_reader_round_trip(tmp_path):
    """stable_ts_amd.model.read_hf_checkpoint: a HuggingFace checkpoint directory written by ``save_pretrained`` comes
    back as upstream-named weights that reproduce the HF model's logits on the oracle (and the oracle's own weights)."""
    import json
    from stable_ts_amd.model import read_hf_checkpoint
    from transformers import WhisperForConditionalGeneration
    torch.manual_seed(0)
    m = om.build_model("tiny.en", seed=7, std=0.05)
    hf_base = _hf_model_from(m)
    full = WhisperForConditionalGeneration(hf_base.config).eval()
    full.model.load_state_dict(hf_base.state_dict())
    full.proj_out.weight = full.model.decoder.embed_tokens.weight
    d = tmp_path / "hf"
    full.save_pretrained(str(d), safe_serialization=True)
    (d / "generation_config.json").write_text(json.dumps(dict(alignment_heads=[[2, 1], [3, 4]])))
    dims, sd, heads = read_hf_checkpoint(str(d))
    assert heads == [(2, 1), (3, 4)]
    assert (dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_text_layer, dims.n_vocab) == \
        (m.dims.n_mels, m.dims.n_audio_ctx, m.dims.n_audio_state, m.dims.n_text_layer, m.dims.n_vocab)
    want = m.state_dict()
    assert set(sd) == {k for k in want if k not in ("decoder.mask", "alignment_heads")}
    for k, v in sd.items():
        assert torch.equal(v, want[k]), k
    back = om.Whisper(om.ModelDimensions(**dims.__dict__))
    back.load_state_dict(sd, strict=False)
    back.eval()
    mel = torch.randn(1, dims.n_mels, 3000) * 0.1
    tokens = torch.tensor([[50257, 50362, 11, 22, 33]])
    with torch.no_grad():
        ours = back(mel, tokens)
        theirs = full(input_features=mel, decoder_input_ids=tokens).logits
    assert (ours - theirs).abs().max().item() < 2e-4


@pytest.mark.parametrize("max_initial", [None, 50, 0])
def test_timestamp_rules_match_hf_port(max_initial):
    """oracle ApplyTimestampRules (upstream whisper/decoding.py, the filter the on-device decode loop re-implements) vs
    ``transformers``' WhisperTimeStampLogitsProcessor -- an independent port of the same upstream rule set -- on random
    logits after random (rule-conforming and rule-breaking) token histories, incl. the first sampled position."""
    from types import SimpleNamespace
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                         WhisperTimeStampLogitsProcessor)
    from oracle.whisper.decoding import ApplyTimestampRules, SuppressBlank, SuppressTokens
    from oracle.whisper.tokenizer import get_tokenizer
    tok = get_tokenizer(False, num_languages=99)
    V = tok.timestamp_begin + 1501
    begin = 3
    cfg = SimpleNamespace(no_timestamps_token_id=tok.no_timestamps, eos_token_id=tok.eot, bos_token_id=tok.sot,
                          max_initial_timestamp_index=max_initial)
    hf = WhisperTimeStampLogitsProcessor(cfg, begin_index=begin)
    ours = ApplyTimestampRules(tok, begin, max_initial)
    g = torch.Generator().manual_seed(0 if max_initial is None else max_initial + 1)
    tb = tok.timestamp_begin
    for trial in range(60):
        n = int(torch.randint(0, 9, (1,), generator=g))
        rows = []
        for _ in range(3):
            hist, last_ts = [], tb
            for i in range(n):
                if torch.rand(1, generator=g) < 0.45:
                    last_ts = min(tb + 1500, last_ts + int(torch.randint(0, 40, (1,), generator=g)))
                    hist.append(last_ts)
                else:
                    hist.append(int(torch.randint(0, tok.eot, (1,), generator=g)))
            rows.append([tok.sot, tok.sot + 1, tok.sot + 2][:begin] + hist)
        tokens = torch.tensor(rows)
        logits = torch.randn(3, V, generator=g) * 3
        logits[:, tb:] += float(torch.randn(1, generator=g)) * 4            # sometimes the timestamp mass wins, sometimes not
        want = hf(tokens, logits.clone())
        got = logits.clone()
        ours.apply(got, tokens)
        assert torch.equal(torch.isinf(got), torch.isinf(want)), trial
        assert torch.equal(got[~torch.isinf(got)], want[~torch.isinf(want)])
    # the two simple filters next to their HF counterparts
    sup = [1, 5, 77, tok.sot, tok.no_speech]
    lg = torch.randn(2, V, generator=g)
    a = lg.clone()
    SuppressTokens(sup).apply(a, torch.zeros(2, 4, dtype=torch.long))
    assert torch.equal(a, SuppressTokensLogitsProcessor(sup)(torch.zeros(2, 4, dtype=torch.long), lg.clone()))
    blank = tok.encode(" ") + [tok.eot]
    for length in (begin, begin + 1):
        a = lg.clone()
        SuppressBlank(tok, begin).apply(a, torch.zeros(2, length, dtype=torch.long))
        assert torch.equal(a, SuppressTokensAtBeginLogitsProcessor(blank, begin)(torch.zeros(2, length, dtype=torch.long), lg.clone()))
