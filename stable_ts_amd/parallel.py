"""Multi-GPU execution: one process per GPU, torch.distributed ("nccl" == RCCL on ROCm, xGMI links inside a node).

The reference is single-process / single-device, so this is a new execution mode: the model is
3 GB on a 288 GB part, hence pure data parallelism over 30-s windows / spans -- no tensor or pipeline split, and NO
collective on the data path.  Two collectives exist in total:
  * start-up: the packed weight arena is broadcast from rank 0 (one ncclBroadcast of ~3.1 GB for large-v3 fp16);
  * end: the per-window result records (a few KB of segments/words per window) are gathered to rank 0.
"""
import os
from typing import Any, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun) and join the process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("SWX_FORCE_DIST") == "1"     # exercise the RCCL code path with a single rank (tests)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_windows(n_windows: int, rank: int, world: int) -> range:
    """Contiguous block partition of window indices (contiguous spans keep the per-rank result order == time order).
    The first (n_windows % world) ranks take one extra window."""
    base, extra = divmod(n_windows, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def broadcast_arena(arena: torch.Tensor, src: int = 0):
    """Weights: rank `src` loads/converts the checkpoint once, everyone else receives the packed arena over RCCL."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("SWX_FORCE_DIST") == "1"):
        dist.broadcast(arena, src=src)
    return arena


def gather_results(local: List[Any], dst: int = 0) -> Optional[List[Any]]:
    """Gather per-rank lists of picklable window records; returns the rank-ordered concatenation on `dst`."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and os.environ.get("SWX_FORCE_DIST") != "1"):
        return list(local)
    world = dist.get_world_size()
    out = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(local, out, dst=dst)
    if dist.get_rank() != dst:
        return None
    return [r for part in out for r in part]


def barrier():
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("SWX_FORCE_DIST") == "1"):
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized() or (dist.get_world_size() == 1 and os.environ.get("SWX_FORCE_DIST") != "1"):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


_QUEUE_SERIAL = [0]


def _span_queue(n_items: int, step: int):
    """Yields the first index of every chunk of `step` items this rank pulls from a shared queue: an atomic fetch-add on a
    counter in the default process group's store (TCPStore).  Every call of transcribe_sharded uses a fresh key (all ranks
    call it the same number of times)."""
    from torch.distributed.distributed_c10d import _get_default_store
    store = _get_default_store()
    _QUEUE_SERIAL[0] += 1
    key = f"swx_span_queue_{_QUEUE_SERIAL[0]}"
    while True:
        first = store.add(key, step) - step
        if first >= n_items:
            return
        yield first


def _first_live_window(audio: torch.Tensor, *, suppress_silence=True, q_levels=20, k_size=5, min_word_dur=None,
                       min_silence_dur=None, nonspeech_skip=None, suppress_ts_tokens=False):
    """The audio of the first fixed-stride 30-s window that transcribe(batch_size=N) would decode: windows the silence analysis
    marks silent are skipped, a window that opens with a non-speech section >= nonspeech_skip is skipped, a later long section
    cuts the window short (transcribe.py `window_input`, original_whisper.py:505-526)."""
    import numpy as np
    from .audio import N_SAMPLES, SAMPLE_RATE
    from .stabilization import NonSpeechPredictor
    mwd = 0.1 if min_word_dur is None else min_word_dur
    # the same predictor transcribe()'s `new_track` builds: with suppress_silence=False the is_silent test counts exact-zero
    # samples, with suppress_ts_tokens it is made on the per-unit mask -- the window the language is settled on must be the one the
    # single-GPU run settles it on
    predictor = NonSpeechPredictor(q_levels=q_levels, k_size=k_size, min_word_dur=mwd, min_silence_dur=min_silence_dur,
                                   get_mask=suppress_ts_tokens, loudness=bool(suppress_silence))
    for k in range(0, int(audio.shape[-1]), N_SAMPLES):
        seg = audio[k:k + N_SAMPLES]
        if not seg.numel():
            continue
        pred = predictor.predict(seg, offset=k / SAMPLE_RATE)
        if pred["is_silent"]:
            continue
        if nonspeech_skip and pred["timings"] is not None:
            starts = pred["timings"][0] - k / SAMPLE_RATE
            ends = pred["timings"][1] - k / SAMPLE_RATE
            long_idx = np.flatnonzero((ends - starts) >= nonspeech_skip)
            if len(long_idx):
                j = long_idx[0]
                if starts[j] < mwd or int(starts[j] * SAMPLE_RATE) == 0:
                    continue
                seg = seg[: int(starts[j] * SAMPLE_RATE)]
        if seg.numel():
            return seg
    return None


def transcribe_sharded(model, audio: torch.Tensor, *, batch_size: int = 8, mode: str = "windows", spans_per_rank: int = 4,
                       work_queue: bool = True, lockstep: int = 2, **kw):
    """Transcription of ONE long recording over all ranks; the segments are gathered on rank 0 (returns a WhisperResult
    there, None elsewhere).

    ``mode="windows"``: rank r takes the contiguous block of 30-s windows `shard_windows(...)` and runs
    `model.transcribe(block, batch_size=...)` (fixed stride, no prompt carry-over).
    ``mode="spans"``: the recording is cut at quiet places into ``world * spans_per_rank`` spans (every rank computes the
    same plan from the same audio); the ranks pull them ``lockstep`` at a time from a shared work queue (``work_queue=False``:
    static contiguous runs) and advance each group in lockstep with the reference's sequential algorithm per span
    (spans.py) -- equal to the reference run once per span, whichever rank ran it."""
    from .audio import N_SAMPLES, SAMPLE_RATE
    from .result import WhisperResult
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if mode == "spans":
        from .spans import merge_span_results, plan_spans, transcribe_spans
        plan = plan_spans(audio.detach().float().cpu(), world * spans_per_rank, q_levels=kw.get("q_levels", 20),
                          k_size=kw.get("k_size", 5))
        parts = []
        if work_queue and world > 1:
            # over-decomposed work queue (speech density varies, a static split leaves ranks idle): the spans
            # are handed out `lockstep` at a time through an atomic counter in the process group's key-value store -- no
            # collective on the data path; every rank keeps pulling until the plan is exhausted
            for first in _span_queue(len(plan), lockstep):
                mine = plan[first:first + lockstep]
                res = transcribe_spans(model, audio, spans=mine, **kw)
                parts.append((int(mine[0][0]), res.to_dict()))          # keyed by the group's start: the queue hands spans out in any order
        else:
            mine = [plan[i] for i in shard_windows(len(plan), rank, world)]
            if mine:
                res = transcribe_spans(model, audio, spans=mine, **kw)
                parts = [(int(mine[0][0]), res.to_dict())]
        gathered = gather_results(parts)
        if gathered is None:
            return None
        # the per-group results are already in recording time: order them by their start, no further shift
        gathered.sort(key=lambda p: p[0])
        return merge_span_results([(0, WhisperResult(d, check_sorted=False)) for _, d in gathered], kw.get("language"))
    if mode != "windows":
        raise ValueError(f"unknown mode {mode!r}")
    kw = dict(kw)
    regroup = kw.pop("regroup", True)
    if not kw.get("language") and getattr(model, "is_multilingual", False):
        # ONE language for the whole recording, settled where the single-GPU window-parallel run settles it (transcribe.py
        # `settle_language`, original_whisper.py:505-532): at the first window the silence analysis does not skip, on that
        # window's audio as trimmed by nonspeech_skip.  Every rank runs the same analysis on the same audio, so all ranks
        # decode with the same tokenizer without a collective.
        first = _first_live_window(audio, suppress_silence=kw.get("suppress_silence", True), q_levels=kw.get("q_levels", 20),
                                   k_size=kw.get("k_size", 5), min_word_dur=kw.get("min_word_dur"),
                                   min_silence_dur=kw.get("min_silence_dur"), nonspeech_skip=kw.get("nonspeech_skip"),
                                   suppress_ts_tokens=kw.get("suppress_ts_tokens", False))
        if first is not None:
            _, probs = model.detect_language(model.log_mel(first, N_SAMPLES - int(first.shape[-1])))
            kw["language"] = max(probs, key=probs.get)
    n_win = (int(audio.shape[-1]) + N_SAMPLES - 1) // N_SAMPLES
    mine = shard_windows(n_win, rank, world)
    rec = dict(segments=[], nonspeech=[], language=kw.get("language"))
    if len(mine):
        span = audio[mine.start * N_SAMPLES: mine.stop * N_SAMPLES]
        off = mine.start * N_SAMPLES / SAMPLE_RATE
        # per rank WITHOUT regrouping: segments may merge / split across shard boundaries, so that runs once on the whole result
        res = model.transcribe(span, batch_size=batch_size, regroup=False, **kw)
        res.offset_time(off)
        rec["segments"] = [s.to_dict() for s in res.segments]
        rec["nonspeech"] = [(d["start"] + off, d["end"] + off) for d in res.nonspeech_sections]
        rec["language"] = res.language or rec["language"]
    parts = gather_results([rec])
    if parts is None:
        return None
    language = next((p["language"] for p in parts if p["language"]), None)
    out = WhisperResult(dict(segments=[s for p in parts for s in p["segments"]], language=language), check_sorted=False)
    sections = [ns for p in parts for ns in p["nonspeech"]]
    if sections:
        out.nonspeech_sections = [dict(start=a, end=b) for a, b in sections]
    if regroup and kw.get("word_timestamps", True):
        from .regroup import regroup_default
        regroup_default(out, regroup)
    return out
