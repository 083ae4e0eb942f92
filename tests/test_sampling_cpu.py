"""Host half of the sample-exact `temperature > 0` path (Engine.decode(torch_rng=True)), checked without a GPU:
  * upstream's Categorical(logits / T).sample() IS argmax(logits / T - log q) with q = ONE exponential_() call of torch's
    generator on the logits' shape (what the selection kernel computes from swx_decode_cfg.noise);
  * engine.reference_loop_iterations() = the number of sampling draws upstream's decoding loop makes, derived from the lengths of
    the sequences that came out (the device loop polls for completion only every few steps, so its own step count overshoots);
  * torch's generator offset (CUDA generators: get_offset / set_offset) is what the engine rewinds -- present in this torch."""
import numpy as np
import pytest
import torch


def test_categorical_sample_is_argmax_of_logits_minus_log_exponential():
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(5, 51866, generator=g) * 3
    logits[:, 100:200] = -np.inf                                   # suppressed tokens
    for T in (0.2, 0.4, 0.8, 1.0):
        for seed in range(10):
            torch.manual_seed(seed)
            want = torch.distributions.Categorical(logits=logits / T).sample()
            torch.manual_seed(seed)
            q = torch.empty(5, 51866).exponential_()
            got = torch.argmax(logits * (1.0 / T) - torch.log(q), dim=-1)
            assert torch.equal(want, got), (T, seed)


@pytest.mark.parametrize("temperature,best_of,sample_len", [(0.4, 5, 24), (1.0, 3, 12), (0.8, 1, 40), (0.6, 2, 3)])
def test_reference_loop_iterations_counts_the_draws_of_upstreams_loop(temperature, best_of, sample_len):
    from stable_ts_amd.engine import reference_loop_iterations
    from oracle import stable as ost
    from oracle.whisper import decoding as od
    from oracle.whisper import model as om
    from oracle.whisper.decoding import DecodingOptions as ODO
    ref = om.build_model("tiny.en", seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5)
    mel = torch.randn(80, 3000, generator=torch.Generator().manual_seed(5)) * 0.3
    draws, rows = [0], []

    class Counting(torch.distributions.Categorical):
        def sample(self, *a, **k):
            draws[0] += 1
            return super().sample(*a, **k)

    real_cat, real_fin = od.Categorical, od.GreedyDecoder.finalize

    def finalize(self, tokens, sum_logprobs):
        rows.append(tokens.clone())                                # [n_audio, n_group, length] before the EOT padding
        return real_fin(self, tokens, sum_logprobs)
    od.Categorical, od.GreedyDecoder.finalize = Counting, finalize
    try:
        for seed in range(4):
            torch.manual_seed(seed)
            draws[0] = 0
            del rows[:]
            task = ost.DecodingTaskStable(ref, ODO(fp16=False, language="en", max_initial_timestamp=None, sample_len=sample_len,
                                                   temperature=temperature, best_of=best_of if best_of > 1 else None))
            task.run(mel.unsqueeze(0))
            toks = rows[0].reshape(-1, rows[0].shape[-1]).numpy()
            n_init, eot = task.sample_begin, task.tokenizer.eot
            lens = [n_init + (list(r[n_init:]).index(eot) if eot in r[n_init:] else len(r) - n_init) for r in toks]
            assert reference_loop_iterations(np.asarray(lens), n_init, sample_len, ref.dims.n_text_ctx) == draws[0], (seed, lens, draws[0])
    finally:
        od.Categorical, od.GreedyDecoder.finalize = real_cat, real_fin


def test_reference_loop_iterations_bounds():
    from stable_ts_amd.engine import reference_loop_iterations as it
    assert it(np.array([[4, 4, 4]]), 4, 224, 448) == 1             # every sequence drew EOT at once
    assert it(np.array([[4, 9, 6]]), 4, 224, 448) == 6             # the longest drew its EOT at iteration 5
    assert it(np.array([[28]]), 4, 24, 448) == 24                  # never ended: sample_len iterations
    assert it(np.array([[448]]), 440, 224, 448) == 9               # context full: tokens.shape[-1] > n_ctx after 9 iterations


def test_cuda_generator_offsets_are_available():
    g = torch.Generator()
    assert hasattr(g, "get_offset") and hasattr(g, "set_offset")
