"""Golden fixtures for the regrouping layer: run the REFERENCE's own ``WhisperResult`` (stable_whisper/result.py) on
seeded synthetic word-timed results with a list of regroup programs, and store inputs + outputs.

Only runs where /root/reference exists (this container); tests/golden/regroup_cases.json.gz is committed and is what
tests/test_regroup_cpu.py compares stable_ts_amd against on machines without the reference.

    python tests/golden/make_regroup_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

ALGOS = [
    "da",
    "cm_sp=.* /。/?/？_sg=.5_sp=,* /，++++50_sl=70_cm",
    "sg=.3_mg=.15+3",
    "sp=./?/!+1_sl=40+6+0",
    "sl=30++1+1",
    "sl=+4_mg=.2++25+1",
    "ms_sd=4.5_cm=2++1",
    "ms_sd=3+0+0+1_mp=,/;+8",
    "sp=,* +0+1_sg=.4+1+1",
    "l=the+ing_sp=./,_mg=1+++0+1+1_us_sl=20",
    "isp_sp=.* _p=.2+.3_cm=+1.5",
    "sg=.2+1_sl=25+++++1",
    "sp=,/.++++++1_mp=.+6++1_p=.1+.1+2+60+1",
    "isp=0_sd=2+1+0+0+1+0+1_csl",
    "rs=0+1+0_rw=0,0+1+0_sg=.25",
    "rp=2+0+1++1+0_sg=.4",
    "rws= the/ and/ Mr.+0+1++0.6++0_mg=.3",
    "ag=.5+1_cm",
    "ag=.9_sp=./?",
    "co=probability+<+0.3+remove+1",
    "co=word+end+,+splitright+1_co=duration+>+1.5+lock+1_sl=20",
    "co=len=word+>+8+mergeright+1_co=text+start+ the+remove+0",
    "co=word+in+any= the, and, fox+splitleft+1",
]

VOCAB = ["the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "and", "then", "running", "singing", "Mr.",
         "U.S.", "Dr.", "3.", "A.", "hello", "world", "this", "is", "a", "longer", "sentence", "with", "several",
         "words", "in", "it", "Thing", "Go", "ok", "是", "的。", "好，", "extraordinarily", "I", "we're", "don't"]
TAILS = ["", "", "", "", "", ",", ",", ".", ".", "?", "!", ";", "...", "。", "，", "？"]


def synth_result(seed: int) -> dict:
    """A transcript-like result: 3-9 segments of 1-30 words, gaps, a few very long words, abbreviations."""
    rng = random.Random(seed)
    t = rng.uniform(0.0, 2.0)
    segs = []
    tok = 100
    for si in range(rng.randint(3, 9)):
        words = []
        for wi in range(rng.choice([1, 2, 3, 5, 8, 13, 21, 30])):
            text = rng.choice(VOCAB)
            if not text.endswith((".", "。", "，")):
                text += rng.choice(TAILS)
            if rng.random() < 0.9:
                text = " " + text
            if rng.random() < 0.05:
                text = " ," + text.strip()
            dur = rng.choice([0.0, 0.08, 0.2, 0.3, 0.45, 0.6, 1.4, 3.2]) * rng.uniform(0.7, 1.3)
            gap = rng.choice([0.0, 0.0, 0.0, 0.05, 0.12, 0.26, 0.5, 0.51, 0.9, 2.5]) if wi else rng.choice([0.0, 0.3, 1.0])
            start = t + gap
            end = start + dur
            ntok = rng.randint(1, 3)
            words.append(dict(word=text, start=round(start, 3), end=round(end, 3),
                              probability=round(rng.random(), 4), tokens=list(range(tok, tok + ntok))))
            tok += ntok
            t = end
        segs.append(dict(start=words[0]["start"], end=words[-1]["end"], text="".join(w["word"] for w in words),
                         seek=round(30.0 * (si // 3), 3), tokens=[x for w in words for x in w["tokens"]],
                         temperature=rng.choice([0.0, 0.2, None]), avg_logprob=-rng.random(),
                         compression_ratio=1.0 + rng.random(), no_speech_prob=rng.random() * 0.2, words=words))
    ns, t0 = [], 0.0
    while t0 < t:                                  # detected non-speech sections (for adjust_gaps)
        t0 += rng.uniform(0.5, 6.0)
        d = rng.choice([0.1, 0.3, 0.8, 2.0])
        ns.append(dict(start=round(t0, 3), end=round(t0 + d, 3)))
        t0 += d
    return dict(language="en", text="".join(s["text"] for s in segs), segments=segs, nonspeech_sections=ns)


def snapshot(res) -> dict:
    """What is compared: per segment its decode statistics and per word text/start/end/locks/tokens."""
    segs = []
    for s in res.segments:
        d = dict(start=s.start, end=s.end, text=s.text, seek=s.seek, temperature=s.temperature,
                 avg_logprob=s.avg_logprob, compression_ratio=s.compression_ratio, no_speech_prob=s.no_speech_prob,
                 tokens=list(s.tokens), id=s.id)
        d["words"] = None if s.words is None else [
            [w.word, w.start, w.end, w.probability, list(w.tokens or []), bool(w.left_locked), bool(w.right_locked),
             w.id, w.segment_id] for w in s.words]
        segs.append(d)
    return dict(segments=segs, history=res.regroup_history, text=res.text)


def main():
    from make_golden import import_reference
    sw = import_reference()
    import contextlib
    import copy
    import io
    cases, inputs = [], {}
    for seed in range(24):
        inp = inputs[str(seed)] = synth_result(seed)
        for algo in ([ALGOS[0]] + [ALGOS[1 + (seed * 5 + k) % (len(ALGOS) - 1)] for k in range(5)]):
            res = sw.WhisperResult(copy.deepcopy(inp))
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    res.regroup(algo)
                cases.append(dict(seed=seed, algo=algo, out=snapshot(res)))
            except Exception as e:                 # some programs make the reference itself fail on some inputs
                cases.append(dict(seed=seed, algo=algo, error=type(e).__name__))
    import gzip
    out = os.path.join(HERE, "regroup_cases.json.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(json.dumps(dict(inputs=inputs, cases=cases), ensure_ascii=False, separators=(",", ":")).encode("utf-8"))
    print(f"wrote {len(cases)} cases ({sum(1 for c in cases if 'error' in c)} raising) -> {out} ({os.path.getsize(out) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
