#!/usr/bin/env python
"""bench.py -- real-time factor of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Default mode (``--mode transcribe``): one "step" = one full pass of model.transcribe() over BASELINE.json configs[2]:
  large-v3 architecture (random-init weights; no checkpoints exist offline), 10 min of synthetic 16 kHz audio
  (20 x 30-s windows, already resident in HBM when the timed region starts), word_timestamps=True, beam_size=5,
  window-parallel batches, fixed decode budget per window (sample_len = min_tokens = 112: random weights have no
  meaningful EOT -- BASELINE.md section 3), temperature 0 without fallback thresholds.  The random weights are shaped
  (stable_ts_amd.BENCH_WEIGHTS: token-embedding gain 9, cross-attention score gain 8, LayerNorm jitter 0.1, timestamp rows
  x0.1 -- the SAME recipe tests/test_gpu_f16_depth.py holds fp16 to the north-star tolerances on, at full depth, 112 steps,
  and tests/test_gpu_batch_invariance.py chains the 20-window batch to) so that every window's transcript keeps ~111 TEXT tokens: the
  word-timestamp stage (teacher-forced scoring pass, attention weights, DTW) then runs at the length a real transcript
  has (reference: timing.py:202-306 sees ~100-225 tokens per window); ``config.words`` / ``config.text_tokens_per_window``
  report what the timed pass produced.
``--mode align`` = BASELINE.json configs[3]: model.align() of a given random token text on 30 min of audio (sequential
  window state machine of the reference, one window per device pass).
``--mode sharded`` = BASELINE.json configs[4]'s shape: ONE recording of world x minutes scattered by 30-s window over the ranks
  (parallel.transcribe_sharded), segments gathered and regrouped on rank 0 inside the timed pass.
``--model base.en --minutes 0.5 --batch 1 --beam 1`` = configs[1] (one window, greedy): ``latency_ms_per_window``.
``--minutes 60 --batch 120`` = one GPU's share of the 8-h / 8-GPU job at the batch size that amortises the decode-step launches
  (at any N; the DEFAULT per-GPU workload is configs[2] at every N, so that the per-N values form one weak-scaling curve).
``--spans K`` = the reference-exact sequential algorithm on K spans in lockstep (spans.py); ``--sequential`` = the default
  transcribe() (one window per pass).  ``--debug-flags`` = the library's A/B switches (16384 no graph replay, 32768 no prefetch).
Weak scaling: every rank processes its own recording; value = N * K * seconds / max-over-ranks wall time.  The default
regrouping is part of the timed pass; ``config.decode_loop`` reports how the decode loops ran (graph captures / replays / eager).

The JSON line also carries
  roofline     -- the dominant kernel class of an extra, instrumented pass (HIP events on the launch stream inside
                  libswx): algorithmic flops (or bytes) / measured time vs the gfx950 peak
  cpu_baseline -- the CPU oracle (oracle/, a port: the reference's CPU path is not runnable offline) on ONE window of the
                  same workload shape (same beam, same number of decode steps -- the first step, which contains the one-off
                  cross-K/V projection, is timed separately from the per-step cost -- and the same number of text tokens
                  through scoring / DTW), timed on this box's host cores (rank 0, N=1 only)
  strict_f32   -- the same pass in the strict-parity mode (dtype f32: exact-f32 MFMA, the mode the bit-exact tests run in)
Diagnostics (extra passes, never part of `value`): ``--phase-times`` (a synchronise per stage: wall ms per stage),
``--host-profile FILE`` (cProfile of one pass), ``--sequential`` (the default transcribe(): one window per pass).
Host threads: the container of the MI355X boxes has a CPU quota (16 CPUs under 256 visible hardware threads); the product's
entry points park torch's intra-op pool, and the CPU baseline here runs on as many threads as the quota allows.
"""
import argparse
import ctypes
import json
import os
import signal
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F16_TFLOPS = 2500.0     # dense fp16/bf16 MFMA, MI355X_MICROARCH.md
PEAK_MFMA_F32_TFLOPS = 157.3      # exact-f32 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0             # HBM3E spec, MI355X_MICROARCH.md
# libswx's profiler classes (csrc/swx_kernels.h SwxProfClass) -> the kernels that dominate each class on this workload
CLASS_NAMES = ["gemm_f16_tiled", "decode-step GEMM", "attn_flash_f16", "attn_decode_cross_f16",
               "self_attn (decode step)", "decode_select", "mel", "align_weights", "dtw", "splitk_finish / layernorm"]
CLASS_BOUND = ["mfma", "hbm", "mfma", "hbm", "hbm", "hbm", "hbm", "hbm", "hbm", "hbm"]

# HBM traffic per launch of the kernel classes from the committed counter run (rocprofv3 --pmc FETCH_SIZE and --pmc
# WRITE_SIZE in separate kernel-trace-only passes; FETCH_SIZE doubled: gfx950 under-reports wide coalesced reads by 2x,
# MI355X_MICROARCH.md).  profiles/pmc_traffic.json is regenerated by scripts/gpu_pmc.sh; not measured per bench run.
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")

LARGE_V3_HEADS = [(l, h) for l, h in ((7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6))]


def synth_audio(seconds: float, seed: int = 0) -> torch.Tensor:
    n = int(seconds * 16000)
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 16000.0
    x = torch.zeros(n)
    for k in range(5):
        f = 120.0 * (k + 1) * (1.0 + 0.37 * k)
        am = 0.5 + 0.5 * torch.sin(2 * np.pi * (1.3 + 0.7 * k) * t + k)
        x += (0.25 / (k + 1)) * torch.sin(2 * np.pi * f * t) * am
    x += 0.01 * torch.randn(n, generator=g)
    for i, s in enumerate(torch.arange(4.0, seconds, 5.0).tolist()):
        d = 0.3 + 0.7 * ((i * 7919) % 10) / 10.0
        x[int(s * 16000): int((s + d) * 16000)] = 0.0
    return (x * 0.5 / x.abs().max()).float()


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries exactly one JSON line)"""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def _pmc_traffic():
    try:
        with open(PMC_FILE) as f:
            return json.load(f)
    except Exception:
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default="transcribe", choices=["transcribe", "align", "sharded"])
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--minutes", type=float, default=None, help="audio per GPU (default 10 for transcribe, 30 for align)")
    ap.add_argument("--batch", type=int, default=None, help="windows per GPU batch (default 20 at every N)")
    ap.add_argument("--host-audio", action="store_true", help="hand transcribe() the recording as a HOST tensor, as the reference's "
                    "callers do (the default keeps it resident in HBM, as the bench contract asks): the PCIe-inclusive rate")
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=112, help="fixed decode budget per window")
    ap.add_argument("--dtype", default="f16")
    # random-weight transcript shape (see the module docstring); --embed-gain 2 --ts-gain 0.5 is the round-1 workload whose
    # transcripts were almost only timestamp tokens (~1 word per window)
    # (defaults = stable_ts_amd.BENCH_WEIGHTS, the recipe the full-depth fp16 parity tests assert on: tests/test_gpu_f16_depth.py)
    ap.add_argument("--embed-gain", type=float, default=9.0)
    ap.add_argument("--ts-gain", type=float, default=0.1)
    ap.add_argument("--ln-jitter", type=float, default=0.1)
    ap.add_argument("--xattn-gain", type=float, default=8.0)
    ap.add_argument("--max-instant-words", type=float, default=1.0)
    ap.add_argument("--align-tokens-per-min", type=int, default=150, help="align mode: text tokens per minute of audio")
    ap.add_argument("--streams", type=int, default=1, help="experimental: host threads / HIP streams per batch (engine clones)")
    ap.add_argument("--spans", type=int, default=0, help="span-parallel mode (the reference's sequential algorithm on this "
                    "many spans in lockstep, spans.py) instead of the fixed-stride window batches")
    ap.add_argument("--sequential", action="store_true", help="the default model.transcribe(audio): the reference's sequential "
                    "window loop (seek from the last timestamp, prompt carried over), one window per device pass")
    ap.add_argument("--raw-seek", action="store_true", help="sequential / span modes: leave every timestamp token selectable.  By default "
                    "these modes suppress the timestamp tokens of 0.02 .. 27.98 s through the reference's own `suppress_tokens` option, "
                    "so that a window's last timestamp -- which the reference's seek advances by (original_whisper.py:629-633) -- lands "
                    "at 28-30 s as it does on speech (~21 windows per 10 min); with random weights it is uniform over 0-30 s (63 windows)")
    ap.add_argument("--no-regroup", action="store_true", help="leave the default regrouping out of the timed pass (round-2 behaviour)")
    ap.add_argument("--debug-flags", type=int, default=0, help="swx_debug_flags() A/B switches (csrc/swx_kernels.h), e.g. 16384 = "
                    "decode loop without the captured step graph")
    ap.add_argument("--ab-flags", type=int, default=0, help="A/B inside ONE process (GPU boxes differ by up to 40 %): after the timed "
                    "steps, three more passes each with and without these swx_debug_flags() bits, alternating; reported as `ab`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the strict-f32 leg")
    ap.add_argument("--host-profile", default="", help="diagnostic: one extra pass under cProfile, the 40 heaviest functions "
                    "(by own time) written to this file; not part of `value`")
    ap.add_argument("--phase-times", action="store_true", help="one extra pass with a device synchronise at every stage "
                    "boundary: wall ms per stage (host share of the pass) under \"phase_ms\"; not part of `value`")
    ap.add_argument("--cpu-budget", type=float, default=240.0, help="hard cap (s) on the CPU-baseline leg")
    args = ap.parse_args()
    # Every N runs the SAME per-GPU workload -- BASELINE.json configs[2]: 10 min, 20 windows in one batch, per rank -- so that the
    # values at N = 1, 2, 4, 8 are one weak-scaling curve (per-GPU work fixed as N grows; rounds 3-5 switched N > 1 to 60 min at
    # 120 windows per batch, which made value(N) / (N x value(1)) compare two workloads).  One GPU's share of configs[4] (8 h over
    # 8 GPUs = 60 min per GPU, at the batch size that amortises the decode-step launches: ~2 020x per GPU against ~1 460x at 20) is
    # `--minutes 60 --batch 120`, at any N.
    if args.minutes is None:
        args.minutes = 30.0 if args.mode == "align" else 10.0
    if args.batch is None:
        args.batch = 20

    import stable_ts_amd as sw
    from stable_ts_amd import parallel as par
    rank, local, world = par.init_from_env()
    # host threads of this process (weight generation on rank 0, the CPU baseline): no more than its share of the container's
    # CPU quota -- torch sizes its pool by the visible hardware threads (256 on the MI355X boxes, under a 16-CPU quota), and
    # OpenMP workers beyond the quota only get every rank throttled
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q_, p_ = f.read().split()[:2]
        if q_ != "max":
            torch.set_num_threads(max(1, min(torch.get_num_threads(), int(int(q_) / int(p_)) // max(world, 1))))
    except Exception:
        pass
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)

    dims = sw.dims_for(args.model)
    heads = LARGE_V3_HEADS if dims.n_text_layer == 32 and dims.n_text_head == 20 else None
    n_win_batch = args.batch if args.mode in ("transcribe", "sharded") else 1

    def make_model(dtype):
        return sw.Whisper(dims, device=dev, dtype=dtype, alignment_heads=heads, max_windows=n_win_batch,
                          max_rows=n_win_batch * (args.beam if args.mode in ("transcribe", "sharded") else 1))

    model = make_model(args.dtype)
    if args.debug_flags:
        model.engine.lib.swx_debug_flags(args.debug_flags)
    sd = None
    if rank == 0:
        log("generating random weights")
        sd = sw.random_state_dict(dims, seed=1234, std=0.02, embed_gain=args.embed_gain, ts_gain=args.ts_gain,
                                  ln_jitter=args.ln_jitter, xattn_gain=args.xattn_gain)
        log("loading weights into the arena")
        model.load_state_dict(sd)
    par.broadcast_arena(model.engine.arena, src=0)        # RCCL broadcast of the packed weights (no-op at N=1)
    if rank != 0:
        model.engine.mark_weights_loaded()

    seconds = args.minutes * 60.0
    if args.mode == "sharded":
        # BASELINE.json configs[4]: ONE recording of world x minutes, every rank holds it (the reference hands transcribe() one
        # file), the 30-s windows are scattered over the ranks by parallel.transcribe_sharded, the segments gathered and
        # regrouped on rank 0.  The recording is a 10-minute synthetic clip tiled (an 8-h synthesis per rank would dominate
        # the start-up); weak scaling: the recording grows with the number of ranks.
        base = synth_audio(min(seconds, 600.0), seed=0)
        reps = int(np.ceil(seconds * world / (base.shape[0] / 16000.0)))
        audio = base.repeat(reps)[: int(seconds * world * 16000)]
        if not args.host_audio:
            audio = audio.to(dev)
    else:
        audio = synth_audio(seconds, seed=rank)
        if not args.host_audio:
            audio = audio.to(dev)
    if args.mode in ("transcribe", "sharded"):
        # regroup (the reference's default 'da' post-processing, result.py) is inside the timed pass since round 3
        kw = dict(language="en", temperature=0.0, logprob_threshold=None, compression_ratio_threshold=None,
                  no_speech_threshold=None, beam_size=args.beam if args.beam > 1 else None, sample_len=args.tokens,
                  min_tokens=args.tokens, word_timestamps=True, regroup=not args.no_regroup, batch_size=args.batch,
                  max_instant_words=args.max_instant_words)
        if args.streams > 1:
            kw["streams"] = args.streams
        if args.spans > 0 or args.sequential:
            kw.pop("batch_size")
            kw.pop("streams", None)
            if not args.raw_seek:
                from stable_ts_amd.tokenizer import get_tokenizer
                tb = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language="en", task="transcribe").timestamp_begin
                kw["suppress_tokens"] = [-1] + list(range(tb + 1, tb + 1400))
        text_tokens = None
    else:
        g = torch.Generator().manual_seed(7 + rank)
        text_tokens = torch.randint(18, 50000, (int(args.align_tokens_per_min * args.minutes),), generator=g).tolist()
        kw = dict(language="en", token_step=100)

    def step(m):
        if args.mode == "align":
            return m.align(audio, list(text_tokens), **kw)
        if args.mode == "sharded":
            return par.transcribe_sharded(m, audio, mode="windows", **kw)
        if args.spans > 0:
            return m.transcribe_spans(audio, min(args.spans, args.batch), **kw)
        return m.transcribe(audio, **kw)

    res = None
    log("warmup")
    for _ in range(args.warmup):
        res = step(model)
    log("timed steps")
    par.barrier()
    torch.cuda.synchronize()
    enc0 = (getattr(model.engine, "encode_calls", 0), getattr(model.engine, "encode_windows", 0))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(model)
    torch.cuda.synchronize()
    par.barrier()
    import torch.distributed as _dist
    # RCCL reduces device tensors; the CPU dry run of this plumbing (tests/test_bench_dist_cpu.py, gloo) reduces on the host
    dt = par.max_over_ranks(time.perf_counter() - t0,
                            device=dev if _dist.is_initialized() and _dist.get_backend() == "nccl" else None)

    log(f"timed region done: {dt:.3f}s for {args.steps} steps")
    ab = None
    if args.ab_flags and world == 1:
        lib_ = model.engine.lib
        ab = {"flags": args.ab_flags, "off_ms": [], "on_ms": []}
        for _ in range(3):
            for name, fl in (("off_ms", args.debug_flags & ~args.ab_flags), ("on_ms", args.debug_flags | args.ab_flags)):
                lib_.swx_debug_flags(fl)
                torch.cuda.synchronize()
                t_ab = time.perf_counter()
                step(model)
                torch.cuda.synchronize()
                ab[name].append(round(1000.0 * (time.perf_counter() - t_ab), 2))
        lib_.swx_debug_flags(args.debug_flags)
        ab["off_min_ms"], ab["on_min_ms"] = min(ab["off_ms"]), min(ab["on_ms"])
    graph_stats = model.engine.graph_stats()
    n_words = len(res.all_words()) if res is not None else 0
    n_segs = len(res.segments) if res is not None else 0
    n_windows = int(np.ceil(seconds / 30.0))
    if args.mode == "sharded" and res is not None:        # rank 0 holds the gathered result of the WHOLE recording
        n_windows *= world
    eot = 50257 if dims.n_vocab >= 51865 else 50256
    n_text = sum(1 for s in (res.segments if res is not None else []) for t in s.tokens if t < eot)
    last_end = max((s.end for s in res.segments), default=0.0) if res is not None else 0.0
    gathered = par.gather_results([dict(rank=rank, segments=n_segs, words=n_words)])

    out = None
    if rank == 0:
        value = world * args.steps * seconds / dt
        if args.mode == "align":
            wl = (f"{args.model} (random-init), align() of {len(text_tokens)} random text tokens on {args.minutes:g} min synthetic "
                  f"16 kHz audio per GPU, token_step=100 (reference's sequential window state machine)")
            metric = f"real-time factor (audio-sec/wall-sec) {args.model} align()"
        else:
            recipe = dict(embed_gain=args.embed_gain, ts_gain=args.ts_gain, ln_jitter=args.ln_jitter, xattn_gain=args.xattn_gain)
            wl = (f"{args.model} (random-init, weight recipe "
                  + ("BENCH_WEIGHTS = the parity-pinned recipe of tests/test_gpu_f16_depth.py" if recipe == sw.BENCH_WEIGHTS else str(recipe))
                  + f"), {args.minutes:g} min synthetic 16 kHz audio per GPU, "
                  f"word_timestamps=True, beam_size={args.beam}, {args.tokens} decode steps/window, "
                  + (f"span-parallel, {min(args.spans, args.batch)} spans in lockstep" + ("" if args.raw_seek else ", seek advance 28-30 s (suppress_tokens)")
                     if args.spans > 0
                     else "sequential windows (reference control flow)" + ("" if args.raw_seek else ", seek advance 28-30 s per window (timestamp tokens of 0.02-27.98 s suppressed through suppress_tokens)")
                     if args.sequential
                     else f"window-parallel batch {args.batch}" + (f", {args.streams} streams" if args.streams > 1 else ""))
                  + (", default regrouping inside the timed pass" if kw.get("regroup") else ", regroup=False")
                  + (", recording handed over as a HOST tensor (PCIe-inclusive)" if args.host_audio else "")
                  + (f"; ONE recording of {args.minutes * world:g} min scattered by 30-s window over {world} rank(s) "
                     f"(parallel.transcribe_sharded), segments gathered + regrouped on rank 0" if args.mode == "sharded" else ""))
            metric = f"real-time factor (audio-sec/wall-sec) {args.model} word_timestamps=True"
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "x real time", "n_gpus": world, "n_gpus_measured": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", **({"debug_flags": args.debug_flags} if args.debug_flags else {}), **({"ab": ab} if ab else {}),
            "latency_ms_per_window": round(1000.0 * dt / args.steps / max(n_windows, 1), 3),
            "config": {"workload": wl, "mode": "spans" if args.spans > 0 else ("sequential" if args.sequential else args.mode),
                       "windows_per_gpu": n_windows, "segments": n_segs, "words": n_words,
                       # what the device actually ran per pass: encoder calls and the 30-s windows in them (align(): the
                       # reference's state machine re-seeks by the last aligned word, so 30 min are ~80 calls, not 60)
                       "encoder_calls_per_pass": round((getattr(model.engine, "encode_calls", 0) - enc0[0]) / args.steps, 1),
                       "encoder_windows_per_pass": round((getattr(model.engine, "encode_windows", 0) - enc0[1]) / args.steps, 1),
                       "text_tokens_per_window": round(n_text / max(n_windows, 1), 1),
                       "last_segment_end_s": round(float(last_end), 2),
                       "decode_loop": graph_stats,
                       "parallelism": f"dp{world} (windows sharded, no data-path collective)",
                       "per_rank": gathered},
        }

    if rank == 0 and args.host_profile:
        import cProfile, pstats, io
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        pr.enable()
        step(model)
        torch.cuda.synchronize()
        pr.disable()
        buf_ = io.StringIO()
        pstats.Stats(pr, stream=buf_).sort_stats("tottime").print_stats(40)
        buf2_ = io.StringIO()
        pstats.Stats(pr, stream=buf2_).sort_stats("cumulative").print_stats(45)
        with open(args.host_profile, "w") as f:
            f.write(buf_.getvalue() + "\n\n=========== cumulative\n" + buf2_.getvalue())

    if rank == 0 and args.phase_times:
        from stable_ts_amd import transcribe as _tr
        _tr.PHASE_TIMES = {}
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(model)
        torch.cuda.synchronize()
        tot = time.perf_counter() - t1
        ph = {k: round(1000.0 * v, 2) for k, v in _tr.PHASE_TIMES.items()}
        ph["pass total (with the extra synchronises)"] = round(1000.0 * tot, 2)
        _tr.PHASE_TIMES = None
        out["phase_ms"] = ph

    # ---- roofline: one extra instrumented pass (not part of `value`)
    if rank == 0 and not args.no_roofline:
        log("instrumented pass (roofline)")
        lib = model.engine.lib
        lib.swx_prof_enable(1)
        step(model)
        buf = (ctypes.c_double * (3 * len(CLASS_NAMES)))()
        lib.swx_prof_collect(buf, len(CLASS_NAMES))
        lib.swx_prof_enable(0)
        # An event pair around a 4-11 us kernel measures the kernel plus part of the pair's own cost (10.8 us by events vs
        # 8.25 us in the rocprofv3 summary of the same command in round 1, profiles/README.md).  Nothing is subtracted:
        # `achieved` is the conservative event-based figure; the elapsed time of an EMPTY pair is reported next to it and
        # the rocprofv3 average of the same kernel is in profiles/.
        ev_us = 0.0
        try:
            st = torch.cuda.current_stream()
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(65)]
            for a_, b_ in pairs:
                a_.record(st)
                b_.record(st)
            torch.cuda.synchronize()
            ev_us = 1000.0 * float(np.median([a_.elapsed_time(b_) for a_, b_ in pairs[1:]]))
        except Exception:
            ev_us = 0.0
        rows = []
        for c, name in enumerate(CLASS_NAMES):
            n, ms, work = buf[3 * c], buf[3 * c + 1], buf[3 * c + 2]
            if n > 0:
                rows.append(dict(kernel=name, bound=CLASS_BOUND[c], launches=int(n), total_ms=round(ms, 3),
                                 avg_us=round(1000.0 * ms / n, 2), work=work))
        rows.sort(key=lambda r: -r["total_ms"])
        pmc = _pmc_traffic()

        def rate(r):
            if not r["work"]:
                return 0.0
            return r["work"] / (r["total_ms"] * 1e-3) / (1e12 if r["bound"] == "mfma" else 1e9)
        if rows:
            top = rows[0]
            ach = rate(top)
            if top["bound"] == "mfma":
                peak, unit = (PEAK_MFMA_F16_TFLOPS if args.dtype == "f16" else PEAK_MFMA_F32_TFLOPS), "TFLOP/s"
            else:
                peak, unit = PEAK_HBM_GBS, "GB/s"
            tr = pmc.get(top["kernel"], {})
            out["roofline"] = {"kernel": top["kernel"], "bound": top["bound"], "achieved": round(ach, 2), "peak": peak,
                               "unit": unit, "frac": round(ach / peak, 4),
                               "traffic": tr.get("bytes_per_launch"),
                               "traffic_unit": "bytes per launch (HBM fetch + write)", "traffic_source": tr.get("source"),
                               "algorithmic_bytes_per_launch": (round(top["work"] / top["launches"]) if top["bound"] == "hbm" else None),
                               "avg_launch_us": top["avg_us"], "empty_event_pair_us": round(ev_us, 2),
                               "launches": top["launches"]}
            out["kernel_time_ms"] = {r["kernel"]: r["total_ms"] for r in rows}
            for r in rows[1:5]:
                pk = PEAK_HBM_GBS if r["bound"] == "hbm" else (PEAK_MFMA_F16_TFLOPS if args.dtype == "f16" else PEAK_MFMA_F32_TFLOPS)
                out.setdefault("roofline_others", []).append(
                    dict(kernel=r["kernel"], bound=r["bound"], achieved=round(rate(r), 2), frac=round(rate(r) / pk, 4),
                         avg_launch_us=r["avg_us"], unit="TFLOP/s" if r["bound"] == "mfma" else "GB/s"))

    # ---- strict-parity mode (f32) on the same workload: one warm-up + three timed passes (median), rank 0 at N=1
    if rank == 0 and world == 1 and not args.no_f32 and args.dtype != "f32":
        log("strict f32 leg")
        try:
            del model
            torch.cuda.empty_cache()
            m32 = make_model("f32")
            m32.load_state_dict(sd)
            step(m32)
            torch.cuda.synchronize()
            t32 = []
            for _ in range(3):                   # three timed passes, the median (boxes vary by a few % between passes)
                t0 = time.perf_counter()
                r32 = step(m32)
                torch.cuda.synchronize()
                t32.append(time.perf_counter() - t0)
            d32 = float(np.median(t32))
            out["strict_f32"] = {"value": round(seconds / d32, 2), "unit": "x real time", "ms_per_step": round(1000 * d32, 1),
                                 "steps": 3, "ms_all_steps": [round(1000 * x, 1) for x in t32], "statistic": "median",
                                 "words": len(r32.all_words()) if r32 is not None else 0,
                                 "note": "same workload, dtype f32 (exact-f32 MFMA; the mode the bit-exact parity tests run in)"}
            del m32
            torch.cuda.empty_cache()
        except BaseException as e:
            out["strict_f32"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        model = None

    # ---- CPU baseline (oracle port), bounded sample, rank 0 at N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle port)")
        model = None
        torch.cuda.empty_cache()

        def _alarm(signum, frame):
            raise TimeoutError(f"cpu baseline exceeded its {args.cpu_budget:g}s budget")
        signal.signal(signal.SIGALRM, _alarm)
        signal.alarm(int(args.cpu_budget))
        try:
            n_txt = int(round(n_text / max(n_windows, 1))) if args.mode != "align" else 100
            out["cpu_baseline"] = cpu_baseline(args, sd, dims, max(n_txt, 1))
        except BaseException as e:   # the baseline must never take the bench line down
            out["cpu_baseline"] = {"value": None, "unit": "x real time", "kind": "port", "error": f"{type(e).__name__}: {e}"}
        finally:
            signal.alarm(0)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if _dist.is_initialized():
        _dist.destroy_process_group()


def cpu_baseline(args, sd, dims, n_text_tokens):
    """The CPU oracle (restated reference path, fp32, torch CPU) on ONE 30-s window of the same workload shape, every stage
    timed: mel, encoder, the beam decode (first step -- prompt prefill + the one-off cross-K/V projections the kv-cache
    hooks do -- timed on its own; the per-step cost from a few further steps, scaled to the same number of steps as the GPU
    pass), and the word-timestamp stage on the same number of text tokens the GPU pass kept per window."""
    from oracle import stable as ost
    from oracle.whisper import model as om
    from oracle.whisper.audio import log_mel_spectrogram
    from oracle.whisper.decoding import DecodingOptions
    from oracle.whisper.tokenizer import get_tokenizer
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:        # a container CPU quota (the MI355X boxes: 256 hardware threads visible, 16 CPUs of quota): more OpenMP workers
        with open("/sys/fs/cgroup/cpu.max") as f:      # than that only get the process throttled
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads")
    m = om.Whisper(om.ModelDimensions(**dims.__dict__))
    m.load_state_dict(sd)
    m.eval()
    mask = torch.zeros(dims.n_text_layer, dims.n_text_head, dtype=torch.bool)
    heads = LARGE_V3_HEADS if dims.n_text_layer == 32 else [(l, h) for l in range(dims.n_text_layer // 2, dims.n_text_layer) for h in range(dims.n_text_head)]
    for l, h in heads:
        mask[l, h] = True
    m.set_alignment_heads_mask(mask)
    audio = synth_audio(30.0, seed=0)
    t = {}
    log("cpu: model built; mel")
    t0 = time.perf_counter()
    mel = log_mel_spectrogram(audio, dims.n_mels)
    t["mel"] = time.perf_counter() - t0
    log("cpu: encoder")
    t0 = time.perf_counter()
    with torch.no_grad():
        xa = m.encoder(mel[None])
    t["encoder"] = time.perf_counter() - t0
    tok = get_tokenizer(m.is_multilingual, num_languages=m.num_languages, language="en", task="transcribe")
    g = torch.Generator().manual_seed(0)
    text_tokens = torch.randint(18, 50000, (n_text_tokens,), generator=g).tolist()
    sample = f"mel + encoder"
    if args.mode != "align":
        log(f"cpu: encoder {t['encoder']:.1f}s; decode")
        extra = 3

        def dec(n):
            opts = DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=n,
                                   beam_size=args.beam if args.beam > 1 else None)
            t0 = time.perf_counter()
            ost.decode_stable(m, mel, opts, audio_features=xa, min_tokens=n)
            return time.perf_counter() - t0
        t_first = dec(1)
        t_more = dec(1 + extra)
        per_step = max(t_more - t_first, 0.0) / extra
        t["decode_first_step"] = t_first
        t["decode_per_step"] = per_step
        t["decode"] = t_first + per_step * (args.tokens - 1)
        sample += (f" + beam-{args.beam} decode (first step {t_first:.2f} s incl. cross-K/V projection, then {extra} steps timed "
                   f"-> {per_step:.3f} s/step x {args.tokens - 1})")
    t0 = time.perf_counter()
    ost.find_alignment(m, tok, text_tokens, mel, 480000, audio_features=xa)
    t["word_timestamps"] = time.perf_counter() - t0
    sample += f" + scoring/alignment/DTW of {n_text_tokens} text tokens"
    total = t["mel"] + t["encoder"] + t.get("decode", 0.0) + t["word_timestamps"]
    return {"value": round(30.0 / total, 4), "unit": "x real time", "cores": cores, "kind": "port",
            "sample": f"one 30-s window of the same workload on the CPU oracle (torch fp32, {cores} threads): " + sample,
            "stage_seconds": {k: round(v, 3) for k, v in t.items()}}


if __name__ == "__main__":
    main()
