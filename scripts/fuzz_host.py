"""Random option sets against the reference on CPU: the host side of transcribe() / align() / refine() on the oracle-backed
stand-in (tests/oracle_engine.py) next to the reference's own functions on the same oracle model.  This is how the two
host-logic differences fixed in round 1 were found (tests/test_transcribe_host_cpu.py::test_transcribe_fuzz_regressions).
Needs /root/reference.  The reference runs its encoder inside disable_sdpa() in the alignment flows and with SDPA in
transcribe; the stand-in follows (``manual_attention_encoder``), because the oracle's two attention code paths differ
by ~1e-7 and that is enough to move a DTW path on the near-uniform attention of random weights.

    python scripts/fuzz_host.py transcribe --seed 3 -n 40
    python scripts/fuzz_host.py align --seed 9 -n 45
    python scripts/fuzz_host.py align_words --seed 3 -n 25
    python scripts/fuzz_host.py refine --seed 3 -n 8
"""
import argparse
import copy
import os
import random
import signal
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import torch  # noqa: E402

BASE = dict(temperature=0.0, logprob_threshold=None, compression_ratio_threshold=None, no_speech_threshold=None)
TRANSCRIBE = dict(
    condition_on_previous_text=[True, False], word_timestamps=[True, True, False], regroup=[True, False, "sg=.3_sl=25"],
    suppress_silence=[True, False], suppress_word_ts=[True, False], use_word_position=[True, False], q_levels=[20, 10],
    k_size=[5, 3], min_word_dur=[0.1, 0.2, None], nonspeech_error=[0.1, 0.3], suppress_ts_tokens=[False, True],
    gap_padding=[" ...", None], max_instant_words=[0.5, 0.2, 1.0], avg_prob_threshold=[0.00002], nonspeech_skip=[0.4, 1.0], beam_size=[2, 3],
    initial_prompt=[" aaat aaau"], prefix=[" aaaw"], dynamic_heads=[3, "3,2"], aligner=["legacy", "legacy", "new"],
    min_silence_dur=[0.2], clip_timestamps=[[2.0, 20.0, 26.0]], temperature=[0.0, (0.0, 0.4), (0.0, 0.6, 1.0)],
    compression_ratio_threshold=[None, 2.4, 1.0], logprob_threshold=[None, -1.0, -30.0], no_speech_threshold=[None, 0.6, 0.01],
    best_of=[2], patience=[1.5], suppress_blank=[True, False], without_timestamps=[False, False, True],
    suppress_tokens=["-1", "1,2,19"], length_penalty=[0.5])
ALIGN = dict(
    token_step=[100, 30, 12], original_split=[False, True], word_dur_factor=[2.0, None, 1.0], max_word_dur=[3.0, None, 1.0],
    nonspeech_skip=[5.0, None, 0.5, 1.5], fast_mode=[False, True], failure_threshold=[None, 0.3, 0.9],
    remove_instant_words=[False, True], suppress_silence=[True, False], suppress_word_ts=[True, False], q_levels=[20, 10],
    k_size=[5, 3], min_word_dur=[0.1, 0.2, None], nonspeech_error=[0.1, 0.3], use_word_position=[True, False],
    regroup=[True, False, "sg=.3"], presplit=[True, False], gap_padding=[" ...", None], dynamic_heads=[3], aligner=["legacy", "new"])
ALIGN_WORDS = dict(
    normalize_text=[True, False], inplace=[True, False], suppress_silence=[True, False], suppress_word_ts=[True, False],
    min_word_dur=[0.1, 0.2], nonspeech_error=[0.1, 0.3], use_word_position=[True, False], regroup=[True, False], q_levels=[20, 10],
    k_size=[5, 3], min_silence_dur=[None, 0.2], presplit=[True, False], gap_padding=[" ...", None], word_dur_factor=[None, None, 2.0])
REFINE = dict(
    steps=[None, "s", "e", "se"], rel_prob_decrease=[0.03, 0.1], abs_prob_decrease=[0.05, 0.01], rel_rel_prob_decrease=[None, 0.1],
    prob_threshold=[0.5, 0.05], rel_dur_change=[0.5, None, 0.2], abs_dur_change=[None, 0.3], word_level=[True, False],
    precision=[0.5, 0.2], single_batch=[False, True])
WORDS = [" aaat", " aaau", " aaax", " aabc", " aabd", " aabg", " aacc", " aadd"]


class _Stall(Exception):
    pass


def _on_alarm(signum, frame):
    raise _Stall()


def words_of(r):
    if r is None or not r.has_words:            # the reference's all_words() raises on segment-level results
        return None if r is None else []
    return [(w.word, w.start, w.end, float(w.probability), list(w.tokens or [])) for w in r.all_words()]


def close(a, b, rel):
    if a is None or b is None:
        return a is b
    return len(a) == len(b) and all(x[:3] == y[:3] and x[4] == y[4] and abs(x[3] - y[3]) <= rel * abs(y[3]) + 1e-12 for x, y in zip(a, b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["transcribe", "align", "align_words", "refine"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("-n", type=int, default=20)
    ap.add_argument("--only", type=int, default=None, help="replay the option stream but run only this iteration")
    ap.add_argument("--timeout", type=int, default=180, help="seconds per call; a reference call that does not return (its "
                    "window loop can stall, DESIGN.md section 7) is reported as STALL and not counted")
    args = ap.parse_args()
    import make_golden as G
    sw = G.import_reference()
    import stable_whisper
    from oracle.whisper.model import build_model
    from oracle_engine import CpuWhisper
    import stable_ts_amd.alignment as A
    import stable_ts_amd.transcribe as T
    from stable_ts_amd.result import WhisperResult
    T._xkv_select = lambda model, xkv, idx: xkv.select(idx)            # the stand-in's device-buffer helper
    ref = build_model("tiny.en", seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5)
    sw.modify_model(ref)
    mine = CpuWhisper(ref)
    rnd = random.Random(args.seed)
    mine.manual_attention_encoder = args.what in ("align", "align_words")
    pool = dict(transcribe=TRANSCRIBE, align=ALIGN, align_words=ALIGN_WORDS, refine=REFINE)[args.what]
    warnings.simplefilter("ignore")
    signal.signal(signal.SIGALRM, _on_alarm)
    bad = 0
    for it in range(args.n):
        opts = {k: rnd.choice(v) for k, v in pool.items() if rnd.random() < (0.3 if args.what == "transcribe" else 0.4)}
        seed = rnd.randrange(1000)
        if args.what == "transcribe":
            opts = dict(BASE, sample_len=rnd.choice([24, 36, 48]), **opts)       # the default 224 tokens is slow on CPU
            if opts.get("word_timestamps") is False:
                opts.pop("dynamic_heads", None), opts.pop("aligner", None)
            audio = G.synth_audio(rnd.choice([31.0, 47.0, 65.0]), seed=seed)
            run = [lambda: ref.transcribe(audio, language="en", verbose=None, ignore_compatibility=True, **opts),
                   lambda: mine.transcribe(audio, language="en", **opts)]
            rel = 1e-8
        elif args.what == "align":
            text = "".join(rnd.choice(WORDS) + rnd.choice(["", "", "", ".", ",", "?", "\n"]) for _ in range(rnd.randrange(3, 40)))
            audio = G.synth_audio(rnd.choice([8.0, 31.0, 55.0]), seed=seed)
            run = [lambda: ref.align(audio, text, language="en", verbose=None, ignore_compatibility=True, **opts),
                   lambda: A.align(mine, audio, text, language="en", **opts)]
            rel = 1e-8
        elif args.what == "align_words":
            audio = G.synth_audio(rnd.choice([31.0, 62.0]), seed=seed)
            d = ref.transcribe(audio, language="en", verbose=None, ignore_compatibility=True, sample_len=rnd.choice([24, 40]),
                               regroup=rnd.choice([True, False]), **BASE).to_dict()
            as_dicts = rnd.random() < 0.4
            bs = rnd.choice([1, 3, 8])

            def given(cls):
                if as_dicts:
                    return [dict(start=s["start"], end=s["end"], text=s["text"]) for s in d["segments"]]
                return cls(copy.deepcopy(d))
            run = [lambda: ref.align_words(audio, given(stable_whisper.WhisperResult), language="en", verbose=None,
                                           ignore_compatibility=True, **opts),
                   lambda: A.align_words(mine, audio, given(WhisperResult), language="en", batch_size=bs, **opts)]
            rel = 1e-8
        else:
            opts.setdefault("precision", 0.5)
            audio = G.synth_audio(rnd.choice([12.0, 20.0]), seed=seed)
            d = ref.transcribe(audio, language="en", verbose=None, ignore_compatibility=True, sample_len=24, regroup=False, **BASE).to_dict()
            ra, rb = stable_whisper.WhisperResult(copy.deepcopy(d)), WhisperResult(copy.deepcopy(d))
            run = [lambda: ref.refine(audio, ra, verbose=None, **opts), lambda: A.refine(mine, audio, rb, **opts)]
            rel = 1e-8
        if args.only is not None and it != args.only:
            continue
        print("START", it, seed, opts, flush=True)                         # a run that stalls shows its options
        res = []
        for f in run:
            torch.manual_seed(0)
            try:
                signal.alarm(args.timeout)
                r = f()
                signal.alarm(0)
                res.append(("ok", words_of(r), None if r is None else [(s.start, s.end, s.text) for s in r.segments],
                            None if (r is None or rel > 1e-8) else r.to_dict()))
            except _Stall:
                res.append(("stall",))
            except Exception as e:                                         # noqa: BLE001
                signal.alarm(0)
                res.append(("raised", type(e).__name__))
                print("   raised:", type(e).__name__, str(e)[:300], flush=True)
                if args.only is not None:
                    import traceback
                    traceback.print_exc(limit=-5)
        if res[0][0] == "stall":
            print(it, "STALL (reference)", res[1][0], flush=True)
            continue
        if res[0][0] == res[1][0] == "ok":
            same = res[0][2] == res[1][2] and close(res[0][1], res[1][1], rel) and res[0][3] == res[1][3]
        else:
            same = res[0] == res[1]
        print(it, "OK" if same else "DIFF", res[0][0], res[1][0], flush=True)
        bad += not same
    print("differences:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
