"""Engine -- thin Python owner of one swx_model handle (libswx.so) on one GPU.

PyTorch is used for device memory (arena, workspace, I/O tensors) and streams only; every arithmetic step of the hot
path is a libswx call.  Nothing here falls back to torch ops or to the CPU.
"""
import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import SWX_F16, SWX_F32, check, swx_decode_cfg, swx_dims
from .audio import N_FRAMES, N_SAMPLES, hann_window, slaney_mel_filterbank


@dataclass
class ModelDimensions:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def reference_loop_iterations(lens, n_init: int, sample_len: int, n_ctx: int) -> int:
    """How many iterations upstream's ``DecodingTask._main_loop`` makes for a job whose sequences came out with these lengths
    (``lens`` = tokens up to, excluding, the first EOT, initial tokens included): iteration i appends token i to every
    sequence; the loop ends after the first iteration at which every sequence has ended, the context is exceeded
    (``tokens.shape[-1] > n_ctx``, decode.py:60) or ``sample_len`` iterations were made.  One sampling draw per iteration."""
    longest = int(np.max(np.asarray(lens))) - int(n_init)        # a sequence of k sampled tokens drew its EOT at iteration k
    return max(1, min(longest + 1, int(sample_len), int(n_ctx) - int(n_init) + 1))


def _i32arr(vals: Sequence[int]):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


class Engine:
    def __init__(self, dims: ModelDimensions, dtype: str = "f16", device: str = "cuda:0",
                 max_windows: int = 1, max_rows: int = 5, alignment_heads: Optional[Sequence[Tuple[int, int]]] = None):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.dims = dims
        self.dtype_name = dtype
        self.dtype = {"f16": SWX_F16, "f32": SWX_F32}[dtype]
        self.tdtype = torch.float16 if dtype == "f16" else torch.float32
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        cd = swx_dims(**{f: getattr(dims, f) for f, _ in swx_dims._fields_})
        h = ctypes.c_void_p()
        check(self.lib.swx_model_create(ctypes.byref(cd), self.dtype, ctypes.byref(h)), "swx_model_create")
        self.h = h
        nbytes = self.lib.swx_weights_bytes(self.h)
        self.arena = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        check(self.lib.swx_bind_weights(self.h, _ptr(self.arena), nbytes), "swx_bind_weights")
        self._load_constants()
        self._heads = None
        if alignment_heads is not None:
            self.set_alignment_heads(alignment_heads)
        self.max_windows = 0
        self.max_rows = 0
        self.ws = None
        self.encode_calls = 0          # device passes through the encoder / 30-s windows in them (bench.py reports both)
        self.encode_windows = 0
        self.reserve(max_windows, max_rows)

    def clone_shared(self, max_windows: Optional[int] = None, max_rows: Optional[int] = None) -> "Engine":
        """A second engine on the SAME weight arena (no copy) with its own workspace: lets another host thread drive
        another HIP stream through the library concurrently (stable_ts_amd.transcribe, ``streams=``).  Experimental."""
        e = object.__new__(Engine)
        e.lib, e.dims, e.dtype_name, e.dtype, e.tdtype, e.device = self.lib, self.dims, self.dtype_name, self.dtype, self.tdtype, self.device
        cd = swx_dims(**{f: getattr(self.dims, f) for f, _ in swx_dims._fields_})
        h = ctypes.c_void_p()
        check(self.lib.swx_model_create(ctypes.byref(cd), self.dtype, ctypes.byref(h)), "swx_model_create")
        e.h = h
        e.arena = self.arena                     # keeps the tensor alive; the C side shares it without clearing it
        check(self.lib.swx_share_weights(e.h, self.h), "swx_share_weights")
        e._heads = None
        if self._heads is not None:
            e.set_alignment_heads(self._heads)
        e.max_windows, e.max_rows, e.ws = 0, 0, None
        e.encode_calls = e.encode_windows = 0
        e.reserve(self.max_windows if max_windows is None else max_windows, self.max_rows if max_rows is None else max_rows)
        return e

    # ------------------------------------------------------------------ lifetime
    def __del__(self):
        try:
            if getattr(self, "h", None):
                torch.cuda.synchronize(self.device)
                self.lib.swx_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reserve(self, max_windows: int, max_rows: int):
        """(Re)bind a workspace large enough for `max_windows` windows and `max_rows` decoder sequences."""
        max_rows = max(max_rows, max_windows)
        if self.ws is not None and max_windows <= self.max_windows and max_rows <= self.max_rows and \
                self.lib.swx_num_alignment_heads(self.h) == self._ws_heads:
            return
        torch.cuda.synchronize(self.device)
        self.ws = None
        nbytes = self.lib.swx_workspace_bytes(self.h, max_windows, max_rows)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        check(self.lib.swx_bind_workspace(self.h, _ptr(self.ws), nbytes, max_windows, max_rows), "swx_bind_workspace")
        self.max_windows, self.max_rows = max_windows, max_rows
        self._ws_heads = self.lib.swx_num_alignment_heads(self.h)

    # ------------------------------------------------------------------ weights
    def _load_one(self, name: str, t: torch.Tensor):
        t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        check(self.lib.swx_load_tensor(self.h, name.encode(), _ptr(t), t.numel(), self.stream), f"load {name}")
        torch.cuda.current_stream(self.device).synchronize()   # `t` may be freed right after

    def _load_constants(self):
        self._load_one("const.hann", hann_window())
        self._load_one("const.mel_filters", torch.from_numpy(slaney_mel_filterbank(self.dims.n_mels)))

    def mark_weights_loaded(self):
        """The arena was filled from outside (RCCL broadcast of rank 0's packed arena, parallel.broadcast_arena): take
        every tensor as present and run the load-time preparation that ``load_state_dict`` ends with."""
        check(self.lib.swx_weights_mark_loaded(self.h), "swx_weights_mark_loaded")
        check(self.lib.swx_weights_finalize(self.h, self.stream), "swx_weights_finalize")
        torch.cuda.current_stream(self.device).synchronize()

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """`sd` uses the upstream checkpoint keys (encoder.blocks.N.attn.query.weight, ...)."""
        for k, v in sd.items():
            t = v.detach().to(device=self.device, dtype=torch.float32).contiguous()
            rc = self.lib.swx_load_tensor(self.h, k.encode(), _ptr(t), t.numel(), self.stream)
            if rc == -10:
                if strict:
                    raise _lib.SwxError(f"unexpected tensor in state dict: {k}")
                continue
            check(rc, f"load {k}")
            torch.cuda.current_stream(self.device).synchronize()
        if strict and not self.lib.swx_weights_complete(self.h):
            buf = ctypes.create_string_buffer(256)
            missing = []
            i = 0
            while self.lib.swx_missing_tensor(self.h, i, buf, 256) and i < 8:
                missing.append(buf.value.decode())
                i += 1
            raise _lib.SwxError(f"missing tensors in state dict: {missing} ...")
        if self.lib.swx_weights_complete(self.h):
            # load-time preparation that needs every tensor (LayerNorm-folded copies for the fused decode step)
            check(self.lib.swx_weights_finalize(self.h, self.stream), "swx_weights_finalize")
            torch.cuda.current_stream(self.device).synchronize()

    def set_alignment_heads(self, pairs: Sequence[Tuple[int, int]]):
        pairs = [tuple(int(v) for v in p) for p in pairs]
        self._heads = pairs
        flat = [v for p in pairs for v in p]
        check(self.lib.swx_set_alignment_heads(self.h, _i32arr(flat), len(pairs)), "swx_set_alignment_heads")
        self.alignment_heads = [tuple(p) for p in pairs]
        if getattr(self, "ws", None) is not None:
            mw, mr = self.max_windows, self.max_rows
            self.max_windows = self.max_rows = 0
            self.reserve(mw, mr)

    @property
    def n_alignment_heads(self) -> int:
        return self.lib.swx_num_alignment_heads(self.h)

    # ------------------------------------------------------------------ a1 mel
    def log_mel(self, pcm: torch.Tensor, per_item_max: bool = False) -> torch.Tensor:
        """pcm f32 [B, 480000] on this device -> f32 [B, n_mels, 3000]."""
        assert pcm.is_cuda and pcm.dtype == torch.float32 and pcm.shape[-1] == N_SAMPLES and pcm.is_contiguous()
        B = pcm.shape[0]
        self.reserve(max(B, self.max_windows), max(self.max_rows, 1))
        mel = torch.empty(B, self.dims.n_mels, N_FRAMES, dtype=torch.float32, device=self.device)
        check(self.lib.swx_log_mel(self.h, _ptr(pcm), B, _ptr(mel), int(per_item_max), self.stream), "swx_log_mel")
        return mel

    def log_mel_ragged(self, pcm: torch.Tensor, n_valid: Sequence[int], n_total: Sequence[int],
                       per_item_max: bool = True) -> torch.Tensor:
        """pcm f32 [B, 480000] (row b holds n_valid[b] samples) -> pad_or_trim(log_mel(segment, padding=n_total-n_valid),
        3000): upstream's frames for a segment that is not padded to 30 s (refine / locate)."""
        assert pcm.is_cuda and pcm.dtype == torch.float32 and pcm.shape[-1] == N_SAMPLES and pcm.is_contiguous()
        B = pcm.shape[0]
        assert len(n_valid) == B and len(n_total) == B
        self.reserve(max(B, self.max_windows), max(self.max_rows, 1))
        mel = torch.empty(B, self.dims.n_mels, N_FRAMES, dtype=torch.float32, device=self.device)
        check(self.lib.swx_log_mel_ragged(self.h, _ptr(pcm), _i32arr(n_valid), _i32arr(n_total), B, _ptr(mel),
                                          int(per_item_max), self.stream),
              "swx_log_mel_ragged")
        return mel

    # ------------------------------------------------------------------ a2 encoder
    def encode(self, mel: torch.Tensor) -> torch.Tensor:
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous()
        if mel.ndim == 2:
            mel = mel[None]
        B = mel.shape[0]
        assert mel.shape[1:] == (self.dims.n_mels, N_FRAMES), mel.shape
        self.reserve(max(B, self.max_windows), max(self.max_rows, 1))
        xa = torch.empty(B, self.dims.n_audio_ctx, self.dims.n_audio_state, dtype=self.tdtype, device=self.device)
        check(self.lib.swx_encode(self.h, _ptr(mel), B, _ptr(xa), self.stream), "swx_encode")
        self.encode_calls += 1
        self.encode_windows += B
        return xa

    def cross_kv(self, xa: torch.Tensor) -> torch.Tensor:
        B = xa.shape[0]
        assert xa.dtype == self.tdtype and xa.is_contiguous()
        nbytes = self.lib.swx_cross_kv_bytes(self.h, B)
        xkv = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        check(self.lib.swx_cross_kv(self.h, _ptr(xa), B, _ptr(xkv), self.stream), "swx_cross_kv")
        xkv.n_windows = B
        return xkv

    # ------------------------------------------------------------------ a3/a4 decode
    def decode(self, xkv: torch.Tensor, init_tokens: Sequence[Sequence[int]], *, n_group: int = 1, beam: bool = False,
               temperature: float = 0.0, patience: Optional[float] = None, sample_len: int = 224, sot_index: int = 0,
               suppress_blank: bool = True, apply_timestamp_rules: bool = True,
               max_initial_timestamp_index: Optional[int] = None, eot: int = 0, sot: int = 0, no_timestamps: int = -1,
               timestamp_begin: int = 0, no_speech: int = -1, blank_token: int = -1,
               suppress_tokens: Sequence[int] = (), ts_mask: Optional[torch.Tensor] = None, min_tokens: int = 0,
               seed: int = 0, window_uid: Optional[Sequence[int]] = None, torch_rng: bool = False):
        """``torch_rng`` (sampling decoder, ``temperature > 0``): draw from torch's generator of this device exactly as the
        reference's loop does -- upstream ``GreedyDecoder.update`` samples ``Categorical(logits / T)``, which is
        ``argmax(p / q)`` with ``q = empty_like(p).exponential_()``: ONE generator call per step on ``[W * G, n_vocab]``.  The
        variates of every possible step are drawn up front (the same calls, so the same numbers), the device loop reads
        them (``swx_decode_cfg.noise``), and the generator is left where the reference leaves it: after as many draws as its
        loop makes iterations.  With the same seed the sampled tokens are then the reference's (up to f32 near-ties of
        ``p / q``).  False: the counter-based hash keyed on ``(seed, window_uid)``, which does not depend on the batch."""
        W = len(init_tokens)
        n_init = len(init_tokens[0])
        assert all(len(t) == n_init for t in init_tokens), "all windows of a job share the initial length"
        self.reserve(max(W, self.max_windows), max(W * n_group, self.max_rows))
        bad = [t for t in suppress_tokens if not 0 <= int(t) < self.dims.n_vocab]
        if bad:       # upstream's SuppressTokens indexes the logits with them: IndexError
            raise IndexError(f"suppress_tokens ids out of range for a vocabulary of {self.dims.n_vocab}: {bad[:8]}")
        uid = None
        if window_uid is not None:
            assert len(window_uid) == W
            uid = _i32arr([int(u) & 0x7FFFFFFF for u in window_uid])
        noise = rng = None
        if torch_rng and temperature > 0 and not beam:
            noise, rng = self._draw_noise(int(sample_len), W * n_group)
        cfg = swx_decode_cfg(
            n_windows=W, n_group=n_group, beam=int(beam), temperature=float(temperature),
            patience=float(patience or 0.0), sample_len=int(sample_len), sample_begin=n_init, sot_index=int(sot_index),
            suppress_blank=int(suppress_blank), apply_timestamp_rules=int(apply_timestamp_rules),
            max_initial_timestamp_index=-1 if max_initial_timestamp_index is None else int(max_initial_timestamp_index),
            eot=eot, sot=sot, no_timestamps=no_timestamps, timestamp_begin=timestamp_begin, no_speech=no_speech,
            blank_token=blank_token, n_suppress=len(suppress_tokens), min_tokens=int(min_tokens), seed=int(seed),
            window_uid=uid, noise=_ptr(noise))
        g_out = self.lib.swx_decode_gout(ctypes.byref(cfg))
        TS = self.dims.n_text_ctx + 1
        d_init = torch.tensor(np.asarray(init_tokens, dtype=np.int32), device=self.device)
        d_sup = torch.tensor(np.asarray(list(suppress_tokens) or [0], dtype=np.int32), device=self.device)
        d_mask = None
        if ts_mask is not None:
            d_mask = ts_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            assert d_mask.shape == (W, 1501)
        toks = torch.empty(W, g_out, TS, dtype=torch.int32, device=self.device)
        lens = torch.empty(W, g_out, dtype=torch.int32, device=self.device)
        sumlp = torch.empty(W, g_out, dtype=torch.float32, device=self.device)
        nosp = torch.empty(W, dtype=torch.float32, device=self.device)
        steps = check(self.lib.swx_decode(self.h, ctypes.byref(cfg), _ptr(d_init), _ptr(d_sup), _ptr(d_mask), _ptr(xkv),
                                          _ptr(toks), _ptr(lens), _ptr(sumlp), _ptr(nosp), self.stream), "swx_decode")
        out = dict(tokens=toks.cpu().numpy(), lens=lens.cpu().numpy(), sum_logprobs=sumlp.cpu().numpy(),
                   no_speech_prob=nosp.cpu().numpy(), steps=steps, sample_begin=n_init)
        if rng is not None:
            gen, off0, inc = rng
            gen.set_offset(off0 + inc * reference_loop_iterations(out["lens"], n_init, int(sample_len), self.dims.n_text_ctx))
            # [sample_len][W * G][n_vocab] f32 = 232 MB for large-v3 at best_of 5: back to torch's caching allocator (the next
            # sampled retry gets the same block without a hipMalloc; other tensors may use it in between)
            self._noise_buf = None
        return out

    def _draw_noise(self, steps: int, rows: int):
        """[steps][rows][n_vocab] Exp(1) variates from torch's generator of this device, one ``exponential_()`` call per step
        (what ``torch.multinomial(p, 1)`` draws inside the reference's ``Categorical.sample()``), and what is needed to put
        the generator back: (generator, offset before, offset consumed per call)."""
        gen = torch.cuda.default_generators[self.device.index if self.device.index is not None else torch.cuda.current_device()]
        buf = getattr(self, "_noise_buf", None)
        if buf is None or buf.shape[0] < steps or buf.shape[1] != rows:
            buf = self._noise_buf = torch.empty(steps, rows, self.dims.n_vocab, dtype=torch.float32, device=self.device)
        off0 = gen.get_offset()
        inc = 0
        for t in range(steps):
            buf[t].exponential_()
            if t == 0:
                inc = gen.get_offset() - off0
        assert inc > 0 and gen.get_offset() == off0 + inc * steps, "torch generator offsets are not linear in the calls"
        return buf, (gen, off0, inc)

    # ------------------------------------------------------------------ a6/a7 score
    def score(self, xkv: torch.Tensor, tokens: Sequence[Sequence[int]], n_frames: Sequence[int], n_sot: int, eot: int,
              qk_scale: float = 1.0, medfilt_width: int = 7, _defer: bool = False):
        """tokens[w] = [*sot_sequence, no_timestamps, *text_tokens, eot].  Returns (token_probs list, neg_matrix
        device tensor [W, max_n, 1500], T list)."""
        W = len(tokens)
        n_tok = [len(t) for t in tokens]
        max_n = max(n_tok)
        self.reserve(max(W, self.max_windows), max(self.max_rows, 1))
        pad = np.full((W, max_n), eot, dtype=np.int32)
        for w, t in enumerate(tokens):
            pad[w, :len(t)] = t
        d_tok = torch.tensor(pad, device=self.device)
        probs = torch.zeros(W, max_n, dtype=torch.float32, device=self.device)
        neg = torch.zeros(W, max_n, self.dims.n_audio_ctx, dtype=torch.float32, device=self.device)
        check(self.lib.swx_score(self.h, _ptr(d_tok), _i32arr(n_tok), W, max_n, n_sot, eot, _i32arr(n_frames),
                                 float(qk_scale), int(medfilt_width), _ptr(xkv), _ptr(probs), _ptr(neg), self.stream),
              "swx_score")
        T = [n - n_sot - 2 for n in n_tok]
        if _defer:
            # enqueued, not waited for: score_finish() copies the probabilities out.  The token tensor rides along so that its
            # memory is not handed out again while the pass is still reading it
            return probs, neg, T, d_tok
        return self.score_finish((probs, neg, T))

    def score_start(self, *args, **kw):
        """`score` without waiting for the device: the caller does host work (word splitting) under the pass and then calls
        `score_finish` with the returned handle."""
        return self.score(*args, _defer=True, **kw)

    @staticmethod
    def score_finish(handle):
        probs, neg, T = handle[:3]
        p = probs.cpu().numpy()
        return [p[w, :T[w]].astype(np.float64).tolist() for w in range(len(T))], neg, T

    def score_qk(self, xkv: torch.Tensor, tokens: Sequence[Sequence[int]], *, n_sot: int, eot: int, row0: int, n_rows: int):
        """Teacher-forced pass that hands out the raw (pre-softmax) attention scores of this engine's alignment heads
        for token rows ``row0 .. row0 + n_rows - 1``: (token_probs list, f32 device tensor [W, heads, n_rows, 1500]).
        Test / inspection hook: the head-selection variants use ``score_q`` + ``heads_dynamic`` / ``heads_new`` instead."""
        W = len(tokens)
        n_tok = [len(t) for t in tokens]
        max_n = max(n_tok)
        self.reserve(max(W, self.max_windows), max(self.max_rows, 1))
        pad = np.full((W, max_n), eot, dtype=np.int32)
        for w, t in enumerate(tokens):
            pad[w, :len(t)] = t
        d_tok = torch.tensor(pad, device=self.device)
        probs = torch.zeros(W, max_n, dtype=torch.float32, device=self.device)
        qk = torch.empty(W, self.n_alignment_heads, n_rows, self.dims.n_audio_ctx, dtype=torch.float32, device=self.device)
        check(self.lib.swx_score_qk(self.h, _ptr(d_tok), _i32arr(n_tok), W, max_n, n_sot, eot, int(row0), int(n_rows),
                                    _ptr(xkv), _ptr(probs), _ptr(qk), self.stream), "swx_score_qk")
        p = probs.cpu().numpy()
        return [p[w, :n_tok[w] - n_sot - 2].astype(np.float64).tolist() for w in range(W)], qk

    def graph_stats(self) -> dict:
        """how the decode loops of this engine ran: captured step graphs, graph replays (two steps each), eager steps"""
        out = (ctypes.c_int64 * 4)()
        check(self.lib.swx_graph_stats(self.h, out), "swx_graph_stats")
        return dict(captures=int(out[0]), replays=int(out[1]), eager_steps=int(out[2]), fell_back=bool(out[3]))

    # ------------------------------------------------------------------ f4: head-selection variants (swx_headsel.hip)
    def score_q(self, xkv: torch.Tensor, tokens: Sequence[int], *, n_sot: int, eot: int) -> dict:
        """Teacher-forced pass of ONE window that keeps the cross-attention queries of every layer ([L, n, d], compute dtype)
        instead of any head's scores: the head-selection kernels recompute score rows from them and the window's cross-K.
        Returns the state the two calls below take (token probabilities included)."""
        n = len(tokens)
        self.reserve(max(1, self.max_windows), max(self.max_rows, 1))
        d_tok = torch.tensor(np.asarray([list(tokens)], dtype=np.int32), device=self.device)
        probs = torch.zeros(1, n, dtype=torch.float32, device=self.device)
        q = torch.empty(self.lib.swx_qcap_bytes(self.h, n), dtype=torch.uint8, device=self.device)
        check(self.lib.swx_score_q(self.h, _ptr(d_tok), _i32arr([n]), n, n_sot, eot, _ptr(xkv), _ptr(probs), _ptr(q), self.stream),
              "swx_score_q")
        scratch = torch.empty(self.lib.swx_heads_scratch_bytes(self.h, n), dtype=torch.uint8, device=self.device)
        p = probs.cpu().numpy()
        return dict(q=q, n=n, n_sot=n_sot, xkv=xkv, scratch=scratch, probs=p[0, :n - n_sot - 2].astype(np.float64).tolist())

    def heads_dynamic(self, st: dict, n_frames: int, *, count: int, qk_scale: float = 1.0, medfilt_width: int = 7,
                      jump_indices=None) -> torch.Tensor:
        """timing.py:87-112 with ``dynamic_heads``: per text-token row the ``count`` heads of the whole decoder whose attention
        mass lies nearest the row's expected frame, then the default z-normalisation / median / head mean.  Returns the NEGATED
        matrix [T + 1, 1500] (f32, device)."""
        n, n_sot = st["n"], st["n_sot"]
        rows = n - n_sot - 1
        F = int(n_frames)
        ld_f = self.dims.n_audio_ctx
        peaks = None
        if jump_indices is not None:                                     # timing.py:96-98: midpoints of the previous pass's jumps
            j = np.pad(np.asarray(jump_indices), (0, 1), constant_values=F)
            peaks = torch.from_numpy(np.ascontiguousarray(j[:-1] + ((j[1:] - j[:-1]) * 0.5), dtype=np.float64)).to(self.device)
            assert peaks.numel() == rows
        sel = torch.empty(1, count, rows, ld_f, dtype=torch.float32, device=self.device)
        check(self.lib.swx_heads_dynamic(self.h, _ptr(st["q"]), n, n_sot, rows, _ptr(st["xkv"]), F, float(qk_scale), int(count),
                                         _ptr(peaks), _ptr(sel), ld_f, _ptr(st["scratch"]), st["scratch"].numel(), self.stream),
              "swx_heads_dynamic")
        return align_weights(sel, [F], qk_scale=qk_scale, medfilt_width=medfilt_width)[0]

    def heads_new(self, st: dict, n_frames: int, *, qk_scale: float = 1.0, medfilt_width: int = 7, topk: int = 20,
                  w_colnorm: float = 1, w_rownorm: float = 1, w_coverage: float = 0) -> torch.Tensor:
        """timing.py:115-163 (``aligner='new'``): the ``topk`` sharpest heads of the whole decoder, column-normalised and
        averaged.  Returns the NEGATED matrix of the text-token rows [T + 1, 1500] (f32, device)."""
        n, n_sot = st["n"], st["n_sot"]
        rows = n - n_sot - 1
        ld_f = self.dims.n_audio_ctx
        neg = torch.zeros(rows, ld_f, dtype=torch.float32, device=self.device)
        check(self.lib.swx_heads_new(self.h, _ptr(st["q"]), n, n, n_sot, rows, _ptr(st["xkv"]), int(n_frames), float(qk_scale),
                                     int(medfilt_width), int(topk), float(w_colnorm), float(w_rownorm), float(w_coverage),
                                     _ptr(neg), ld_f, _ptr(st["scratch"]), st["scratch"].numel(), self.stream), "swx_heads_new")
        return neg

    def pool_matrices(self, negs: Sequence[torch.Tensor], n_heads: Sequence[int]) -> torch.Tensor:
        """timing.py:177-189 (``extra_models``): the head mean over the heads of several models from each model's own (negated)
        head mean: sum_m (H_m / sum H) * neg_m"""
        assert 1 <= len(negs) <= 8 and all(x.shape == negs[0].shape and x.is_contiguous() for x in negs)
        tot = float(sum(n_heads))
        out = torch.empty_like(negs[0])
        ptrs = (ctypes.c_void_p * len(negs))(*[x.data_ptr() for x in negs])
        coef = (ctypes.c_float * len(negs))(*[h / tot for h in n_heads])
        check(self.lib.swx_weighted_sum(ptrs, coef, len(negs), _ptr(out), out.numel(), self.stream), "swx_weighted_sum")
        return out

    def median_filter(self, x: torch.Tensor, width: int) -> torch.Tensor:
        return median_filter(x, width)

    def forward_logits(self, xkv: torch.Tensor, tokens: Sequence[Sequence[int]], pad_token: int = 0) -> torch.Tensor:
        W = len(tokens)
        n_tok = [len(t) for t in tokens]
        max_n = max(n_tok)
        self.reserve(max(W, self.max_windows), max(self.max_rows, 1))
        pad = np.full((W, max_n), pad_token, dtype=np.int32)
        for w, t in enumerate(tokens):
            pad[w, :len(t)] = t
        d_tok = torch.tensor(pad, device=self.device)
        out = torch.empty(W, max_n, self.dims.n_vocab, dtype=torch.float32, device=self.device)
        check(self.lib.swx_forward_logits(self.h, _ptr(d_tok), _i32arr(n_tok), W, max_n, _ptr(xkv), _ptr(out),
                                          self.stream), "swx_forward_logits")
        return out

    # ------------------------------------------------------------------ a8 dtw
    def dtw(self, x: torch.Tensor, N: Sequence[int], M: Sequence[int]):
        """x f32 device [W, ld_n, ld_m] (the NEGATED alignment matrix); returns [(text_idx, time_idx)] int64 numpy."""
        return dtw(x, N, M)


def dtw(x: torch.Tensor, N: Sequence[int], M: Sequence[int]):
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.ndim == 3
    W, ld_n, ld_m = x.shape
    dev = x.device
    dN = torch.tensor(np.asarray(N, dtype=np.int32), device=dev)
    dM = torch.tensor(np.asarray(M, dtype=np.int32), device=dev)
    cap = ld_n + ld_m
    ti = torch.empty(W, cap, dtype=torch.int32, device=dev)
    tj = torch.empty(W, cap, dtype=torch.int32, device=dev)
    ln = torch.empty(W, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.swx_dtw_workspace_bytes(W, ld_n, ld_m), dtype=torch.uint8, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    check(lib.swx_dtw(_ptr(x), W, ld_n, ld_m, _ptr(dN), _ptr(dM), _ptr(ti), _ptr(tj), _ptr(ln), _ptr(ws), stream), "swx_dtw")
    ti, tj, ln = ti.cpu().numpy(), tj.cpu().numpy(), ln.cpu().numpy()
    return [(ti[w, :ln[w]].astype(np.int64), tj[w, :ln[w]].astype(np.int64)) for w in range(W)]


def loudness_probe(chunks: Sequence[torch.Tensor]) -> list:
    """Device half of the non-VAD silence analysis for a batch of windows that are resident on the GPU
    (``swx_loudness_probe``): per window ``(n, thr, idx, vals)`` -- the k-th largest |x| (k = 0.1 % of the samples,
    nonvad.py:19-22) and |x| at ``stabilization.probe_indices(n)`` -- or None for a window too short for a mask.  One launch,
    one copy-out of 24 KB per window (the full-length host path copies 1.9 MB per window and selects on the host)."""
    from .stabilization import probe_indices
    lib = _lib.load()
    W = len(chunks)
    dev = chunks[0].device
    ns = [int(c.shape[-1]) for c in chunks]
    idxs = [probe_indices(n) for n in ns]
    n_idx = max((len(i) for i in idxs if i is not None), default=0)
    if n_idx == 0:
        return [None] * W
    stride = max(ns)
    buf = torch.empty(W, stride, dtype=torch.float32, device=dev)
    h_idx = np.full((W, n_idx), -1, dtype=np.int32)
    h_nk = np.zeros((W, 2), dtype=np.int32)
    for w, (c, n, ix) in enumerate(zip(chunks, ns, idxs)):
        buf[w, :n] = c.detach().to(dtype=torch.float32)
        h_nk[w] = (n, int(n * 0.001))
        if ix is not None:
            h_idx[w, :len(ix)] = ix
    d_idx = torch.from_numpy(h_idx).to(dev)
    d_nk = torch.from_numpy(h_nk).to(dev)
    out = torch.empty(W, n_idx + 1, dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    scratch = torch.empty(max(int(lib.swx_loudness_probe_scratch_bytes(W)), 16), dtype=torch.uint8, device=dev)
    check(lib.swx_loudness_probe(_ptr(buf), stride, _ptr(d_nk), _ptr(d_idx), n_idx, W, _ptr(out), _ptr(scratch), scratch.numel(),
                                 stream), "swx_loudness_probe")
    h = out.cpu()
    res = []
    for w, (n, ix) in enumerate(zip(ns, idxs)):
        res.append(None if ix is None else (n, float(h[w, 0]), ix, h[w, 1:1 + len(ix)].clone()))
    return res


def median_filter(x: torch.Tensor, width: int) -> torch.Tensor:
    """whisper.timing.median_filter on the device (f32, last axis)."""
    lib = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32
    xc = x.contiguous()
    n = xc.shape[-1]
    rows = xc.numel() // n
    out = torch.empty_like(xc)
    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    # the kernel's grid.y carries the row index: chunk it
    flat_in, flat_out = xc.view(rows, n), out.view(rows, n)
    for r0 in range(0, rows, 65535):
        r1 = min(rows, r0 + 65535)
        check(lib.swx_median_filter(_ptr(flat_in[r0:r1]), r1 - r0, n, width, _ptr(flat_out[r0:r1]), stream), "swx_median_filter")
    return out


def align_weights(qk: torch.Tensor, n_frames: Sequence[int], qk_scale: float = 1.0, medfilt_width: int = 7) -> torch.Tensor:
    """qk f32 device [W, H, N, ld_f] raw scaled attention logits -> NEGATED mean matrix [W, N, ld_f]."""
    lib = _lib.load()
    assert qk.is_cuda and qk.dtype == torch.float32 and qk.is_contiguous() and qk.ndim == 4
    W, H, N, ld_f = qk.shape
    out = torch.zeros(W, N, ld_f, dtype=torch.float32, device=qk.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(qk.device).cuda_stream)
    scratch = torch.empty(max(lib.swx_align_weights_scratch_bytes(W, H, N), 256), dtype=torch.uint8, device=qk.device)
    check(lib.swx_align_weights(_ptr(qk), W, H, N, ld_f, _i32arr(n_frames), float(qk_scale), int(medfilt_width), _ptr(out),
                                _ptr(scratch), scratch.numel(), stream), "swx_align_weights")
    return out
