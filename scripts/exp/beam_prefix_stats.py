"""How much of the self-attention KV cache do the five beams of a window SHARE during the benchmark's decode?  (round 6, VERDICT r5
item 5: a kernel that loads a cache row once per window instead of once per beam only pays if the beams share their ancestors.)
Decodes the bench batch (20 windows x beam 5) truncated at several lengths and counts, per window, the distinct token prefixes per
position among the five live beams = the distinct (cache row, position) pairs the step's self-attention reads.
    python scripts/exp/beam_prefix_stats.py  ->  gpurun_out/r06_beam_prefix_stats.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HEADS = ((7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6))


def main():
    import bench
    import stable_ts_amd as sw
    from stable_ts_amd.decoding import DecodingOptions, DecodingPlan
    W, G, STEPS = 20, 5, 112
    dims = sw.dims_for("large-v3")
    model = sw.Whisper(dims, device="cuda:0", dtype="f16", alignment_heads=HEADS, max_windows=W, max_rows=W * G)
    model.load_state_dict(sw.random_state_dict(dims, seed=1234, std=0.02, **sw.BENCH_WEIGHTS))
    audio = bench.synth_audio(30.0 * W, seed=0).cuda()
    wins = [audio[i * 480000:(i + 1) * 480000].contiguous() for i in range(W)]
    plan = DecodingPlan(model, DecodingOptions(language="en", beam_size=G, sample_len=STEPS, min_tokens=STEPS, max_initial_timestamp=None))
    kw, init = plan.engine_kwargs(), list(plan.initial_tokens)
    xkv = model.cross_kv(model.encoder(model.log_mel_batch(wins, [0] * W)))
    out = {}
    for s in (8, 16, 32, 48, 64, 80, 96, 112):
        o = model.engine.decode(xkv, [init] * W, **dict(kw, sample_len=s))
        sb = o["sample_begin"]
        ratios, full = [], 0
        for w in range(W):
            rows = [tuple(o["tokens"][w, k, sb: sb + int(o["lens"][w, k])].tolist()) for k in range(o["tokens"].shape[1]) if int(o["lens"][w, k]) > 0]
            n = min(len(r) for r in rows)
            uniq = sum(len({r[: j + 1] for r in rows}) for j in range(n))          # distinct prefixes per position
            ratios.append((uniq + len(init)) / float(n + len(init)))              # the initial tokens are one shared row each
            full += uniq == len(rows) * n
        out[s] = dict(mean_unique_rows_per_position=float(np.mean(ratios)), min=float(np.min(ratios)), max=float(np.max(ratios)),
                      per_window=[round(x, 2) for x in ratios])
        print(s, out[s]["mean_unique_rows_per_position"], out[s]["min"], out[s]["max"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_beam_prefix_stats.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
