#!/usr/bin/env python
"""Host-only cost of a transcribe() pass under N-way CPU contention: what the 8 ranks of one node do
to each other on the HOST while every GPU works for its own rank.

    python scripts/host_contention_probe.py [--procs 8] [--passes 5] [--minutes 10] [--batch 20]      (on a GPU box)

1. RECORD (needs the GPU): one pass of the bench workload (large-v3 random-init, window-parallel batches, beam 5, word
   timestamps, default regrouping) with every device call's result copied out: `Engine.decode`, `score_start/finish`, `dtw`,
   `loudness_probe`.  Spectrogram / encoder / cross-K/V hand back opaque buffers the host never reads.
2. REPLAY (host only, no GPU calls): the same `model.transcribe()` host code on a stand-in that returns the recorded results
   in order -- token bookkeeping, segment slicing, word splitting / assembly, silence snapping, result model, regrouping --
   timed as ms per pass: once alone, then `--procs` processes at the same time (each the way a rank runs it: its own
   process, torch's intra-op pool parked by the product's entry point).
Output: one JSON line {host_ms_per_pass_alone, host_ms_per_pass_contended (median / max over the processes), ratio, cpu quota,
device pass ms for scale}.  If the ratio stays near 1 the host side of a rank does not depend on its neighbours.
"""
import argparse
import json
import os
import pickle
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_quota():
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else int(q) / int(p)
    except Exception:
        return None


def transcribe_kwargs(args):
    return dict(language="en", temperature=0.0, logprob_threshold=None, compression_ratio_threshold=None, no_speech_threshold=None,
                beam_size=args.beam, sample_len=args.tokens, min_tokens=args.tokens, word_timestamps=True, batch_size=args.batch,
                max_instant_words=1.0)


# ------------------------------------------------------------------------------------------------------------------ record
def record(args, path):
    import torch
    import stable_ts_amd as sw
    import stable_ts_amd.engine as E
    from bench import LARGE_V3_HEADS, synth_audio
    dims = sw.dims_for(args.model)
    heads = LARGE_V3_HEADS if dims.n_text_layer == 32 and dims.n_text_head == 20 else None
    model = sw.Whisper(dims, device="cuda:0", dtype="f16", alignment_heads=heads, max_windows=args.batch, max_rows=args.batch * args.beam)
    model.load_state_dict(sw.random_state_dict(dims, seed=1234, std=0.02, **sw.BENCH_WEIGHTS))
    audio = synth_audio(args.minutes * 60.0, seed=0).cuda()
    kw = transcribe_kwargs(args)
    model.transcribe(audio, **kw)                                     # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.transcribe(audio, **kw)
    torch.cuda.synchronize()
    device_pass_ms = 1000.0 * (time.perf_counter() - t0)

    tape = dict(decode=[], score=[], dtw=[], probe=[])
    eng = model.engine
    real_decode, real_finish, real_dtw, real_probe = eng.decode, eng.score_finish, eng.dtw, E.loudness_probe

    def decode(*a, **k):
        out = real_decode(*a, **k)
        tape["decode"].append(out)
        return out

    def score_finish(handle):
        probs, neg, T = real_finish(handle)
        tape["score"].append((probs, T, tuple(neg.shape)))
        return probs, neg, T

    def dtw(neg, N, M):
        paths = real_dtw(neg, N, M)
        tape["dtw"].append(paths)
        return paths

    def probe(chunks):
        res = real_probe(chunks)
        tape["probe"].append([None if r is None else (False if r is False else (r[0], r[1], r[2], r[3].cpu())) for r in res])
        return res
    eng.decode, eng.score_finish, eng.dtw, E.loudness_probe = decode, score_finish, dtw, probe
    try:
        res = model.transcribe(audio, **kw)
    finally:
        eng.decode, eng.score_finish, eng.dtw, E.loudness_probe = real_decode, real_finish, real_dtw, real_probe
    with open(path, "wb") as f:
        pickle.dump(dict(tape=tape, audio=None if args.record_only else audio.cpu(), minutes=args.minutes, dims=dims.__dict__, is_multilingual=model.is_multilingual,
                         num_languages=model.num_languages, n_words=len(res.all_words()), kw=kw), f)
    return device_pass_ms, len(res.all_words())


# ------------------------------------------------------------------------------------------------------------------ replay
def replay(path, passes, profile=""):
    """host-only passes on the recorded tape; prints ms per pass"""
    import types
    import torch
    import stable_ts_amd.engine as E
    import stable_ts_amd.transcribe as T
    from stable_ts_amd.engine import ModelDimensions
    with open(path, "rb") as f:
        rec = pickle.load(f)
    tape = rec["tape"]
    if rec.get("audio") is None:                   # a tape made with --record-only leaves the (seeded, synthetic) recording out
        from bench import synth_audio
        rec["audio"] = synth_audio(rec["minutes"] * 60.0, seed=0)

    class DeviceAudio(torch.Tensor):               # the recording's windows were resident on the GPU: the host code asks
        is_cuda = property(lambda self: True)      # `is_cuda` to choose the probe path; nothing else about it is used

        def cpu(self, *a, **k):
            return self.as_subclass(torch.Tensor)

    class Buf:                                     # opaque device buffer (mel / encoder output / cross-K/V)
        def __init__(self, n):
            self.n_windows = n

        def __getitem__(self, idx):
            return Buf(len(idx) if hasattr(idx, "__len__") else 1)

    class ReplayEngine:
        tdtype = torch.float16
        dtype_name = "f16"
        device = torch.device("cpu")

        def __init__(self):
            self.dims = ModelDimensions(**rec["dims"])
            self.pos = dict(decode=0, score=0, dtw=0, probe=0)

        def _next(self, kind):
            i = self.pos[kind]
            self.pos[kind] = i + 1
            return tape[kind][i]

        def decode(self, xkv, init_tokens, **kw):
            return self._next("decode")

        def score_start(self, *a, **kw):
            return None

        def score_finish(self, handle):
            probs, T, shape = self._next("score")
            return probs, Buf(shape[0]), T

        def dtw(self, neg, N, M):
            return self._next("dtw")

    class ReplayWhisper:
        def __init__(self):
            self.engine = ReplayEngine()
            self.dims = self.engine.dims
            self.is_multilingual, self.num_languages = rec["is_multilingual"], rec["num_languages"]
            self.device = torch.device("cpu")
            self.transcribe = types.MethodType(T.transcribe_stable, self)

        def log_mel_batch(self, audios, paddings=None):
            return Buf(len(audios))

        def encoder(self, mel):
            return mel

        def cross_kv(self, xa):
            return xa

    model = ReplayWhisper()
    T._xkv_select = lambda m, xkv, idx: Buf(len(idx))
    E.loudness_probe = lambda chunks: model.engine._next("probe")
    audio = rec["audio"].as_subclass(DeviceAudio) if tape["probe"] else rec["audio"]
    times = []
    pr = None
    for i_ in range(passes + 1):
        if profile and i_ == 1:
            import cProfile
            pr = cProfile.Profile()
            pr.enable()
        model.engine.pos = dict(decode=0, score=0, dtw=0, probe=0)
        t0 = time.perf_counter()
        res = model.transcribe(audio, **rec["kw"])
        times.append(1000.0 * (time.perf_counter() - t0))
        assert len(res.all_words()) == rec["n_words"], (len(res.all_words()), rec["n_words"])
    if pr is not None:
        import io
        import pstats
        pr.disable()
        buf, buf2 = io.StringIO(), io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(45)
        pstats.Stats(pr, stream=buf2).sort_stats("cumulative").print_stats(70)
        with open(profile, "w") as f:
            f.write(f"{passes} replayed passes\n" + buf.getvalue() + "\n\n=========== cumulative\n" + buf2.getvalue())
    print(json.dumps(dict(ms=sorted(times[1:])[len(times[1:]) // 2], all=times[1:])), flush=True)


def run_replays(path, n, passes):
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--replay", path, "--passes", str(passes)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(n)]
    out = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        if p.returncode != 0:
            raise RuntimeError(se[-2000:])
        out.append(json.loads(so.strip().splitlines()[-1])["ms"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--batch", type=int, default=20)
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=112)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--replay", default="")
    ap.add_argument("--tape", default="/tmp/swx_host_tape.pkl")
    ap.add_argument("--record-only", action="store_true", help="write the tape (without the audio) and stop: replay / profile it elsewhere")
    ap.add_argument("--profile", default="", help="with --replay: cProfile of the replayed passes, sorted by own time, written here")
    args = ap.parse_args()
    if args.replay:
        return replay(args.replay, args.passes, args.profile)
    device_pass_ms, n_words = record(args, args.tape)
    if args.record_only:
        print(json.dumps(dict(tape=args.tape, device_pass_ms=round(device_pass_ms, 1), words=n_words)))
        return
    alone = run_replays(args.tape, 1, args.passes)[0]
    many = sorted(run_replays(args.tape, args.procs, args.passes))
    print(json.dumps({"workload": f"{args.model}, {args.minutes:g} min, batch {args.batch}, beam {args.beam}, {n_words} words per pass",
                      "device_pass_ms (1 rank, measured with the GPU)": round(device_pass_ms, 1),
                      "host_ms_per_pass_alone": round(alone, 1),
                      f"host_ms_per_pass_{args.procs}_procs_median": round(many[len(many) // 2], 1),
                      f"host_ms_per_pass_{args.procs}_procs_max": round(many[-1], 1),
                      "ratio_max_over_alone": round(many[-1] / alone, 3),
                      "cpu_quota": cpu_quota(), "visible_cpus": os.cpu_count()}), flush=True)


if __name__ == "__main__":
    main()
