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


def transcribe_sharded(model, audio: torch.Tensor, *, batch_size: int = 8, mode: str = "windows", spans_per_rank: int = 4,
                       **kw):
    """Transcription of ONE long recording over all ranks; the segments are gathered on rank 0 (returns a WhisperResult
    there, None elsewhere).

    ``mode="windows"``: rank r takes the contiguous block of 30-s windows `shard_windows(...)` and runs
    `model.transcribe(block, batch_size=...)` (fixed stride, no prompt carry-over).
    ``mode="spans"``: the recording is cut at quiet places into ``world * spans_per_rank`` spans (every rank computes the
    same plan from the same audio), rank r takes a contiguous run of them and advances them in lockstep with the
    reference's sequential algorithm per span (spans.py) -- equal to the reference run once per span."""
    from .audio import N_SAMPLES, SAMPLE_RATE
    from .result import WhisperResult
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if mode == "spans":
        from .spans import merge_span_results, plan_spans, transcribe_spans
        plan = plan_spans(audio.detach().float().cpu(), world * spans_per_rank, q_levels=kw.get("q_levels", 20),
                          k_size=kw.get("k_size", 5))
        mine = [plan[i] for i in shard_windows(len(plan), rank, world)]
        parts = []
        if mine:
            res = transcribe_spans(model, audio, spans=mine, **kw)
            parts = [(0, res.to_dict())]
        gathered = gather_results(parts)
        if gathered is None:
            return None
        return merge_span_results([(o, WhisperResult(d, check_sorted=False)) for o, d in gathered], kw.get("language"))
    if mode != "windows":
        raise ValueError(f"unknown mode {mode!r}")
    n_win = (int(audio.shape[-1]) + N_SAMPLES - 1) // N_SAMPLES
    mine = shard_windows(n_win, rank, world)
    segs = []
    if len(mine):
        span = audio[mine.start * N_SAMPLES: mine.stop * N_SAMPLES]
        res = model.transcribe(span, batch_size=batch_size, **kw)
        res.offset_time(mine.start * N_SAMPLES / SAMPLE_RATE)
        segs = [s.to_dict() for s in res.segments]
    allsegs = gather_results(segs)
    if allsegs is None:
        return None
    return WhisperResult(dict(segments=allsegs, language=kw.get("language")), check_sorted=False)
