"""Batch invariance of the benchmarked configuration: large-v3 at full depth (32 + 32 layers), fp16,
stable_ts_amd.BENCH_WEIGHTS, 20 windows x 5 beams x 112 decode steps -- the launch shapes bench.py times (M = 100 rows in
gemm_dec_f16<3,40,*>, gemm_f16_big8 / attn_flash2_f16<.,4> in the encoder) -- against the SAME window run alone (M = 5 rows,
ring / 128-tile GEMMs, 32 queries per wave), which is the shape tests/test_gpu_f16_depth.py pins to the f32 CPU oracle.
Window k of the batch must equal window k alone: encoder output, decoded tokens of every beam, sum_logprobs, no-speech
probability, token probabilities of the scoring pass, DTW path, word times -- bit for bit (every kernel variant computes an
output element from the same operands in the same order, whatever the batch around it).  That chains the batch-20 figures to
the oracle-checked single-window ones.  Reference: the reference runs one window at a time (original_whisper.py:492-710)."""
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HEADS = ((7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6))
W, G, STEPS = 20, 5, 112
# EVERY window of the batch is compared with the window alone since round 6 (a second or less each: 36 s for both dtypes on hardware,
# profiles/r06_batch_invariance_all20_*.json; rounds 4-5 compared windows 0 / 7 / 19).  SWX_BENCH_WINDOWS="0,7,19" narrows it.
import os
ALONE = tuple(range(W))
if os.environ.get("SWX_BENCH_WINDOWS", "all") != "all":
    ALONE = tuple(int(x) for x in os.environ["SWX_BENCH_WINDOWS"].split(","))


def _synth(seconds, seed):
    import bench
    return bench.synth_audio(seconds, seed=seed)


@pytest.mark.parametrize("dtype", ("f16", "f32"))
def test_window_of_a_20_window_batch_equals_the_window_alone(dtype):
    """f32 (round 6): bench.py's strict_f32 leg is timed at batch 20, where the strict mode's GEMM dispatch
    (gemm_f32_rows64 up to 128 rows, gemm_f32_tiled above) and its attention kernels see other launch shapes than a window alone."""
    import stable_ts_amd as sw
    from stable_ts_amd.decoding import DecodingOptions, DecodingPlan
    from stable_ts_amd.timing import AlignmentJob, find_alignment_batch
    from stable_ts_amd.transcribe import _xkv_select
    dims = sw.dims_for("large-v3")
    model = sw.Whisper(dims, device="cuda:0", dtype=dtype, alignment_heads=HEADS, max_windows=W, max_rows=W * G)
    model.load_state_dict(sw.random_state_dict(dims, seed=1234, std=0.02, **sw.BENCH_WEIGHTS))
    audio = _synth(30.0 * W, 0).cuda()
    wins = [audio[i * 480000:(i + 1) * 480000].contiguous() for i in range(W)]
    opts = DecodingOptions(language="en", beam_size=G, sample_len=STEPS, min_tokens=STEPS, max_initial_timestamp=None)
    plan = DecodingPlan(model, opts)
    kw, init, tok = plan.engine_kwargs(), list(plan.initial_tokens), plan.tokenizer

    def run(idx):
        mel = model.log_mel_batch([wins[i] for i in idx], [0] * len(idx))
        xa = model.encoder(mel)
        xkv = model.cross_kv(xa)
        out = model.engine.decode(xkv, [init] * len(idx), **kw)
        res = plan.results(out, [None] * len(idx), ["en"] * len(idx))
        jobs = [AlignmentJob(tok, [t for t in r.tokens if t < tok.eot], 480000) for r in res]
        words = find_alignment_batch(model, jobs, xkv, return_debug=True)
        return dict(xa=xa.clone(), out=out, res=res, jobs=jobs, words=words)

    full = run(list(range(W)))
    assert full["out"]["steps"] == STEPS
    st = model.engine.graph_stats()
    if dtype == "f16":
        assert st["replays"] >= (STEPS - 4) // 2 and not st["fell_back"], st      # the batch ran from the captured step graph
    n_text = [len(j.text_tokens) for j in full["jobs"]]
    assert min(n_text) >= 100, n_text                                             # a transcript-like token mix in every window
    report = {}
    for k in ALONE:
        one = run([k])
        rep = dict(
            encoder_bit_identical=bool(torch.equal(one["xa"][0], full["xa"][k])),
            encoder_max_abs_diff=float((one["xa"][0].float() - full["xa"][k].float()).abs().max()),
            tokens_identical=bool(np.array_equal(one["out"]["tokens"][0], full["out"]["tokens"][k])),
            lens_identical=bool(np.array_equal(one["out"]["lens"][0], full["out"]["lens"][k])),
            sum_logprobs_bit_identical=bool(np.array_equal(one["out"]["sum_logprobs"][0], full["out"]["sum_logprobs"][k])),
            max_abs_dsumlp=float(np.abs(one["out"]["sum_logprobs"][0] - full["out"]["sum_logprobs"][k]).max()),
            no_speech_bit_identical=bool(one["out"]["no_speech_prob"][0] == full["out"]["no_speech_prob"][k]),
            token_probs_bit_identical=one["jobs"][0].debug["token_probs"] == full["jobs"][k].debug["token_probs"],
            dtw_path_identical=bool(np.array_equal(one["jobs"][0].debug["path"][0], full["jobs"][k].debug["path"][0]) and
                                    np.array_equal(one["jobs"][0].debug["path"][1], full["jobs"][k].debug["path"][1])),
            words_identical=[(w.word, w.start, w.end, w.probability) for w in one["words"][0]] ==
                            [(w.word, w.start, w.end, w.probability) for w in full["words"][k]],
            text_tokens=n_text[k], words=len(full["words"][k]))
        report[k] = rep
        del one
    import json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", f"batch_invariance_report_{dtype}.json"), "w") as f:
        json.dump(report, f, indent=1)
    for k, rep in report.items():
        for key in ("encoder_bit_identical", "tokens_identical", "lens_identical", "sum_logprobs_bit_identical",
                    "no_speech_bit_identical", "token_probs_bit_identical", "dtw_path_identical", "words_identical"):
            assert rep[key], (k, key, rep)
    del model, full
    gc.collect()
    torch.cuda.empty_cache()
