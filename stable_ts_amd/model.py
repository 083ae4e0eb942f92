"""The model object: same attribute surface the reference's glue reads from ``whisper.model.Whisper`` after
``stable_whisper.load_model`` / ``modify_model`` (whisper_word_level/original_whisper.py:931-1009; SURVEY.md 8b seam B1),
backed by an ``Engine`` (libswx.so) instead of torch modules.
"""
import os
import types
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .audio import N_SAMPLES
from .engine import Engine, ModelDimensions

# upstream architecture table (whisper/__init__.py::_MODELS dims; facts)
_DIMS = {
    "tiny.en": (80, 1500, 384, 6, 4, 51864, 448, 384, 6, 4),
    "tiny": (80, 1500, 384, 6, 4, 51865, 448, 384, 6, 4),
    "base.en": (80, 1500, 512, 8, 6, 51864, 448, 512, 8, 6),
    "base": (80, 1500, 512, 8, 6, 51865, 448, 512, 8, 6),
    "small.en": (80, 1500, 768, 12, 12, 51864, 448, 768, 12, 12),
    "small": (80, 1500, 768, 12, 12, 51865, 448, 768, 12, 12),
    "medium.en": (80, 1500, 1024, 16, 24, 51864, 448, 1024, 16, 24),
    "medium": (80, 1500, 1024, 16, 24, 51865, 448, 1024, 16, 24),
    "large-v1": (80, 1500, 1280, 20, 32, 51865, 448, 1280, 20, 32),
    "large-v2": (80, 1500, 1280, 20, 32, 51865, 448, 1280, 20, 32),
    "large-v3": (128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 32),
    "large": (128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 32),
    "large-v3-turbo": (128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 4),
    "turbo": (128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 4),
}


def available_models() -> List[str]:
    return list(_DIMS)


def dims_for(name: str) -> ModelDimensions:
    return ModelDimensions(*_DIMS[name])


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Encoder positional embedding (a buffer in upstream checkpoints; regenerated when a state dict lacks it)."""
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


# The random-weight recipe of the benchmark AND of the full-depth fp16 parity tests (bench.py, tests/test_gpu_f16_depth.py,
# tests/test_gpu_batch_invariance.py): token-embedding gain 9 and cross-attention score gain 8 give the top-1 / top-2 logit gaps
# and the peaky attention maps of a trained model, the LayerNorm jitter gives the affine parameters values other than (1, 0),
# timestamp rows x0.1 keep ~111 text tokens per 112-step window (a transcript-like token mix) while the timestamp logits stay
# distinguishable (x0.01 made the 1 501 timestamp logits equal to within 0.02: the beams of a beam search then differ only in
# a near-tied initial timestamp and fp16 rounding picks another one -- measured in round 4, scripts/f16_error_budget.py).
BENCH_WEIGHTS = dict(embed_gain=9.0, ts_gain=0.1, ln_jitter=0.1, xattn_gain=8.0)


def random_state_dict(dims: ModelDimensions, seed: int = 1234, std: float = 0.02, embed_gain: float = 1.0,
                      ts_gain: float = 1.0, ln_jitter: float = 0.0, xattn_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded random weights under the upstream checkpoint keys (no checkpoint exists offline).  Same generator order
    as the test oracle so that both sides see identical weights for a given seed."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    da, dt = dims.n_audio_state, dims.n_text_state

    def rnd(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    def ln(prefix, d):
        sd[prefix + ".weight"] = torch.ones(d)
        sd[prefix + ".bias"] = torch.zeros(d)

    def attn(prefix, d):
        sd[prefix + ".query.weight"] = rnd(d, d)
        sd[prefix + ".query.bias"] = rnd(d)
        sd[prefix + ".key.weight"] = rnd(d, d)
        sd[prefix + ".value.weight"] = rnd(d, d)
        sd[prefix + ".value.bias"] = rnd(d)
        sd[prefix + ".out.weight"] = rnd(d, d)
        sd[prefix + ".out.bias"] = rnd(d)

    def block(prefix, d, cross):
        attn(prefix + ".attn", d)
        ln(prefix + ".attn_ln", d)
        if cross:
            attn(prefix + ".cross_attn", d)
            ln(prefix + ".cross_attn_ln", d)
        sd[prefix + ".mlp.0.weight"] = rnd(4 * d, d)
        sd[prefix + ".mlp.0.bias"] = rnd(4 * d)
        sd[prefix + ".mlp.2.weight"] = rnd(d, 4 * d)
        sd[prefix + ".mlp.2.bias"] = rnd(d)
        ln(prefix + ".mlp_ln", d)

    # order follows nn.Module.state_dict() of upstream's Whisper (encoder first)
    sd["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, da)
    sd["encoder.conv1.weight"] = rnd(da, dims.n_mels, 3)
    sd["encoder.conv1.bias"] = rnd(da)
    sd["encoder.conv2.weight"] = rnd(da, da, 3)
    sd["encoder.conv2.bias"] = rnd(da)
    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", da, False)
    ln("encoder.ln_post", da)
    sd["decoder.positional_embedding"] = torch.randn(dims.n_text_ctx, dt, generator=g) * 0.01
    sd["decoder.token_embedding.weight"] = torch.randn(dims.n_vocab, dt, generator=g) * std * embed_gain
    if ts_gain != 1.0:   # shrink the 1501 timestamp rows: text tokens win more often -> richer synthetic transcripts
        sd["decoder.token_embedding.weight"][dims.n_vocab - 1501:] *= ts_gain
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", dt, True)
    ln("decoder.ln", dt)
    if ln_jitter:   # non-trivial LayerNorm affine parameters from a generator of their own (the other tensors stay the same)
        g2 = torch.Generator().manual_seed(seed + 7919)
        for k in sd:
            if "_ln" in k or k.endswith("ln.weight") or k.endswith("ln.bias") or "ln_post" in k:
                sd[k] = sd[k] + ln_jitter * torch.randn(sd[k].shape, generator=g2)
    if xattn_gain != 1.0:   # sharper cross-attention (scores scaled by xattn_gain): word timing on peaky attention maps
        for k in sd:
            if ".cross_attn.query.weight" in k or ".cross_attn.key.weight" in k:
                sd[k] = sd[k] * float(xattn_gain) ** 0.5
    return sd


class SparseHeads:
    """Stand-in for the sparse bool tensor ``model.alignment_heads`` (timing.py:105 calls ``.indices().T``)."""

    def __init__(self, pairs: Sequence[Tuple[int, int]]):
        self.pairs = [tuple(p) for p in pairs]

    def indices(self) -> torch.Tensor:
        return torch.tensor(self.pairs, dtype=torch.long).reshape(-1, 2).T

    def __len__(self):
        return len(self.pairs)


class Whisper:
    """MI355X-native Whisper.  ``model.transcribe / align / align_words`` are bound like the reference's
    ``modify_model`` does (original_whisper.py:931-949)."""

    def __init__(self, dims: ModelDimensions, device: Union[str, torch.device] = "cuda:0", dtype: str = "f16",
                 alignment_heads: Optional[Sequence[Tuple[int, int]]] = None, max_windows: int = 1, max_rows: int = 5):
        self.dims = dims
        if alignment_heads is None:   # upstream default: every head of the upper half of the decoder
            alignment_heads = [(l, h) for l in range(dims.n_text_layer // 2, dims.n_text_layer) for h in range(dims.n_text_head)]
        self.engine = Engine(dims, dtype=dtype, device=str(device), max_windows=max_windows, max_rows=max_rows,
                             alignment_heads=alignment_heads)
        self.alignment_heads = SparseHeads(alignment_heads)
        self.dq = False
        self._bind_api()

    @classmethod
    def from_engine(cls, engine: Engine) -> "Whisper":
        """A model object around an engine that already exists (weights loaded, heads set)."""
        self = object.__new__(cls)
        self.dims = engine.dims
        self.engine = engine
        self.alignment_heads = SparseHeads(getattr(engine, "alignment_heads", None) or [])
        self.dq = False
        self._bind_api()
        return self

    # -- reference-visible attributes
    @property
    def device(self) -> torch.device:
        return self.engine.device

    @property
    def is_multilingual(self) -> bool:
        return self.dims.n_vocab >= 51865

    @property
    def num_languages(self) -> int:
        return self.dims.n_vocab - 51765 - int(self.is_multilingual)

    @property
    def dtype(self) -> torch.dtype:
        return self.engine.tdtype

    def set_alignment_heads(self, pairs: Sequence[Tuple[int, int]]):
        self.engine.set_alignment_heads(pairs)
        self.alignment_heads = SparseHeads(pairs)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        if "encoder.positional_embedding" not in sd:
            sd = dict(sd)
            sd["encoder.positional_embedding"] = sinusoids(self.dims.n_audio_ctx, self.dims.n_audio_state)
        self.engine.load_state_dict(sd, strict=strict)
        return self

    # -- hot-path pieces with the upstream call shapes
    def log_mel(self, audio: torch.Tensor, padding: int = 0) -> torch.Tensor:
        """whisper.audio.log_mel_spectrogram for one <=30 s segment (len + padding must be 480000; the callers
        guarantee it: original_whisper.py:528-530, alignment.py:410-413).  Returns f32 [n_mels, 3000] on the device."""
        return self.log_mel_batch([audio], [padding])[0]

    def log_mel_batch(self, audios: Sequence[torch.Tensor], paddings: Optional[Sequence[int]] = None) -> torch.Tensor:
        B = len(audios)
        buf = torch.zeros(B, N_SAMPLES, dtype=torch.float32, device=self.device)
        for b, a in enumerate(audios):
            a = torch.as_tensor(a, dtype=torch.float32)
            n = a.shape[-1]
            pad = 0 if paddings is None else paddings[b]
            if n + pad != N_SAMPLES:
                raise ValueError(f"segment length + padding must be {N_SAMPLES}, got {n} + {pad}")
            buf[b, :n] = a.to(self.device)
        return self.engine.log_mel(buf, per_item_max=True)

    def log_mel_segments(self, audios: Sequence[torch.Tensor], padding: int = 0, batch_max: bool = False) -> torch.Tensor:
        """``pad_or_trim(log_mel_spectrogram(a, n_mels, padding=padding), N_FRAMES)`` for segments of at most 30 s that
        are NOT padded to 30 s first -- refine's inference (alignment.py:660-661) and locate (alignment.py:924-925).
        Returns f32 [B, n_mels, 3000] on the device; the frames past ``(len + padding) // 160`` are 0.0.
        ``batch_max``: the clamp floor comes from the max over the whole batch, as upstream's batched call computes it."""
        B = len(audios)
        buf = torch.zeros(B, N_SAMPLES, dtype=torch.float32, device=self.device)
        n_valid = []
        for b, a in enumerate(audios):
            a = torch.as_tensor(a, dtype=torch.float32)
            n = int(a.shape[-1])
            if n > N_SAMPLES:
                raise ValueError(f"segment longer than {N_SAMPLES} samples: {n}")
            buf[b, :n] = a.to(self.device)
            n_valid.append(n)
        return self.engine.log_mel_ragged(buf, n_valid, [n + int(padding) for n in n_valid],
                                          per_item_max=not batch_max)

    def encoder(self, mel: torch.Tensor) -> torch.Tensor:
        mel = mel.to(device=self.device, dtype=torch.float32)
        if mel.ndim == 2:
            mel = mel[None]
        return self.engine.encode(mel.contiguous())

    embed_audio = encoder

    def cross_kv(self, audio_features: torch.Tensor) -> torch.Tensor:
        return self.engine.cross_kv(audio_features.contiguous())

    def logits(self, tokens: Sequence[Sequence[int]], audio_features: torch.Tensor) -> torch.Tensor:
        """model(mel, tokens) / model.logits: teacher-forced full-sequence logits, f32 [W, n, n_vocab]."""
        return self.engine.forward_logits(self.cross_kv(audio_features), tokens)

    def decoder(self, tokens: torch.Tensor, audio_features: torch.Tensor, kv_cache=None) -> torch.Tensor:
        """``model.decoder(tokens, xa)`` as the reference's glue calls it (timing.py:61, alignment.py:988): tokens int
        [B, n] (tensor or nested list) -> logits f32 [B, n, n_vocab] of a teacher-forced pass.  The incremental KV-cached
        form belongs to the on-device decode loop (``swx_decode``); a ``kv_cache`` dict cannot be honoured here."""
        if kv_cache:
            raise NotImplementedError("kv_cache is internal to the device decode loop (swx_decode); call model.decoder "
                                      "with the full token sequence")
        toks = tokens.tolist() if torch.is_tensor(tokens) else [list(t) for t in tokens]
        if toks and not isinstance(toks[0], list):
            toks = [toks]
        xa = audio_features if audio_features.ndim == 3 else audio_features[None]
        if len(toks) != xa.shape[0]:
            if xa.shape[0] != 1:
                raise ValueError(f"{len(toks)} token rows for {xa.shape[0]} audio windows")
            xa = xa.expand(len(toks), -1, -1)
        return self.logits(toks, xa)

    def __call__(self, mel: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        """``model(mel, tokens)`` (alignment.py:660-667)"""
        return self.decoder(tokens, self.encoder(mel))

    forward = __call__

    def install_kv_cache_hooks(self, cache=None):
        raise NotImplementedError("there are no torch modules to hook: the KV cache lives inside swx_decode "
                                  "(include/swx.h); use model.decoder(tokens, xa) for teacher-forced passes")

    def detect_language(self, mel_or_features: torch.Tensor, tokenizer=None):
        """model.detect_language (original_whisper.py:329): argmax / softmax over the language tokens at <|sot|>."""
        from .tokenizer import get_tokenizer
        if tokenizer is None:
            tokenizer = get_tokenizer(self.is_multilingual, num_languages=self.num_languages)
        if tokenizer.language is None or tokenizer.language_token not in tokenizer.sot_sequence:
            raise ValueError("This model doesn't have language tokens so it can't perform lang id")
        x = mel_or_features
        single = x.ndim == 2
        if single:
            x = x[None]
        if tuple(x.shape[-2:]) != (self.dims.n_audio_ctx, self.dims.n_audio_state):
            x = self.encoder(x)
        lg = self.logits([[tokenizer.sot]] * x.shape[0], x)[:, 0].float().cpu()
        mask = torch.ones(lg.shape[-1], dtype=torch.bool)
        mask[list(tokenizer.all_language_tokens)] = False
        lg[:, mask] = -np.inf
        lang_tokens = lg.argmax(dim=-1)
        probs = lg.softmax(dim=-1)
        out = [{c: probs[i, j].item() for j, c in zip(tokenizer.all_language_tokens, tokenizer.all_language_codes)}
               for i in range(x.shape[0])]
        return (lang_tokens[0], out[0]) if single else (lang_tokens, out)

    def clone_for_stream(self) -> "Whisper":
        """A view of this model for another host thread / HIP stream: shares the weight arena, owns its workspace."""
        import copy
        other = copy.copy(self)
        other.engine = self.engine.clone_shared()
        other._bind_api()
        return other

    def stream_context(self):
        """a fresh side stream that waits for the work already queued on the current stream"""
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(s), s

    def _bind_api(self):
        from .transcribe import transcribe_minimal, transcribe_stable
        from .alignment import align, align_words, refine
        self.transcribe = types.MethodType(transcribe_stable, self)
        self.transcribe_stable = self.transcribe
        self.transcribe_minimal = types.MethodType(transcribe_minimal, self)
        self.align = types.MethodType(align, self)
        self.align_words = types.MethodType(align_words, self)
        self.refine = types.MethodType(refine, self)
        from .locator import locate
        self.locate = types.MethodType(locate, self)
        from .spans import transcribe_spans
        self.transcribe_spans = types.MethodType(transcribe_spans, self)


def _read_checkpoint(path: str):
    # weights_only: a checkpoint is data (upstream's are {"dims": dict, "model_state_dict": tensors}); a pickle that needs
    # arbitrary classes to load is refused rather than executed
    import pickle
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        raise RuntimeError(f"{path}: not a plain-data checkpoint (upstream's are {{'dims': dict, 'model_state_dict': tensors}}); a "
                           f"pickle that needs classes to load -- e.g. 'dims' saved as a ModelDimensions object -- is refused. "
                           f"Re-save it with dims=dataclasses.asdict(dims).  ({e})") from e
    if not isinstance(ckpt, dict) or "dims" not in ckpt or "model_state_dict" not in ckpt:
        raise RuntimeError(f"{path}: expected a checkpoint of the form {{'dims': dict, 'model_state_dict': tensors}}")
    return ModelDimensions(**dict(ckpt["dims"])), ckpt["model_state_dict"]


# HuggingFace parameter names -> upstream checkpoint names (the inverse of the table the reference keeps for its HF back
# end, whisper_word_level/hf_whisper.py:30-51): applied as ordered prefix / infix rewrites
_HF_RENAMES = (
    ("model.", ""), ("encoder.embed_positions.weight", "encoder.positional_embedding"),
    ("decoder.embed_positions.weight", "decoder.positional_embedding"), ("decoder.embed_tokens", "decoder.token_embedding"),
    ("encoder.layer_norm", "encoder.ln_post"), ("decoder.layer_norm", "decoder.ln"), (".layers.", ".blocks."),
    (".encoder_attn_layer_norm", ".cross_attn_ln"), (".encoder_attn.", ".cross_attn."), (".self_attn_layer_norm", ".attn_ln"),
    (".self_attn.", ".attn."), (".final_layer_norm", ".mlp_ln"), (".fc1", ".mlp.0"), (".fc2", ".mlp.2"),
    (".q_proj", ".query"), (".k_proj", ".key"), (".v_proj", ".value"), (".out_proj", ".out"),
)


def read_hf_checkpoint(path: str):
    """A HuggingFace Whisper checkpoint directory (``config.json`` + ``model.safetensors`` / ``pytorch_model.bin``
    [+ ``generation_config.json``]) -> (dims, upstream-named state dict, alignment heads or None).  The projection
    ``proj_out`` is tied to the token embedding upstream and is dropped."""
    import json
    with open(os.path.join(path, "config.json"), "r", encoding="utf-8") as f:
        c = json.load(f)
    dims = ModelDimensions(n_mels=c["num_mel_bins"], n_audio_ctx=c["max_source_positions"], n_audio_state=c["d_model"],
                           n_audio_head=c["encoder_attention_heads"], n_audio_layer=c["encoder_layers"],
                           n_vocab=c["vocab_size"], n_text_ctx=c["max_target_positions"], n_text_state=c["d_model"],
                           n_text_head=c["decoder_attention_heads"], n_text_layer=c["decoder_layers"])
    st = os.path.join(path, "model.safetensors")
    if os.path.isfile(st):
        from safetensors.torch import load_file
        raw = load_file(st)
    elif os.path.isfile(os.path.join(path, "pytorch_model.bin")):
        raw = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    else:
        raise RuntimeError(f"no model.safetensors / pytorch_model.bin under {path} (sharded checkpoints: merge them first)")
    sd = {}
    for k, v in raw.items():
        if k.startswith("proj_out."):
            continue
        for a, b in _HF_RENAMES:
            k = (b + k[len(a):] if k.startswith(a) else k) if a == "model." else k.replace(a, b)
        sd[k] = v
    heads = None
    gen = os.path.join(path, "generation_config.json")
    if os.path.isfile(gen):
        with open(gen, "r", encoding="utf-8") as f:
            heads = json.load(f).get("alignment_heads")
        heads = [tuple(int(x) for x in p) for p in heads] if heads else None
    return dims, sd, heads


# The cross-attention heads upstream marks as time-aligned for its official checkpoints (whisper/__init__.py
# ``_ALIGNMENT_HEADS``, stored there as base85-packed boolean masks; the (layer, head) lists below are the same data as
# published in the checkpoints' generation configs).  Restated from memory -- there is no copy of either source in this
# container to diff against, so re-check this table against upstream before relying on it with real weights.  Models
# that are not listed fall back to upstream's constructor default: every head of the upper half of the decoder.
OFFICIAL_ALIGNMENT_HEADS = {
    "tiny.en": [(1, 0), (2, 0), (2, 5), (3, 0), (3, 1), (3, 2), (3, 3), (3, 4)],
    "tiny": [(2, 2), (3, 0), (3, 2), (3, 3), (3, 4), (3, 5)],
    "base.en": [(3, 3), (4, 7), (5, 1), (5, 5), (5, 7)],
    "base": [(3, 1), (4, 2), (4, 3), (4, 7), (5, 1), (5, 2), (5, 4), (5, 6)],
    "small.en": [(6, 6), (7, 0), (7, 3), (7, 8), (8, 2), (8, 5), (8, 7), (9, 0), (9, 4), (9, 8), (9, 10), (10, 0), (10, 1),
                 (10, 2), (10, 3), (10, 6), (10, 11), (11, 2), (11, 4)],
    "small": [(5, 3), (5, 9), (8, 0), (8, 4), (8, 7), (8, 8), (9, 0), (9, 7), (9, 9), (10, 5)],
    "medium.en": [(11, 4), (14, 1), (14, 12), (14, 14), (15, 4), (16, 0), (16, 4), (16, 9), (17, 12), (17, 14), (18, 7),
                  (18, 10), (18, 15), (20, 0), (20, 3), (20, 9), (20, 14), (21, 12)],
    "medium": [(13, 15), (15, 4), (15, 15), (16, 1), (20, 0), (23, 4)],
    "large-v1": [(9, 19), (11, 2), (11, 4), (11, 17), (22, 7), (22, 11), (22, 17), (23, 2), (23, 15)],
    "large-v2": [(10, 12), (13, 17), (16, 11), (16, 12), (16, 13), (17, 15), (17, 16), (18, 4), (18, 11), (18, 19), (19, 11),
                 (21, 2), (21, 3), (22, 3), (22, 9), (22, 12), (23, 5), (23, 7), (23, 13), (25, 5), (26, 1), (26, 12), (27, 15)],
    "large-v3": [(7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6)],
    "large": [(7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6)],
    "large-v3-turbo": [(2, 4), (2, 11), (3, 3), (3, 6), (3, 11), (3, 14)],
    "turbo": [(2, 4), (2, 11), (3, 3), (3, 6), (3, 11), (3, 14)],
}


def load_model(name: str, device: Optional[Union[str, torch.device]] = None, download_root: str = None,
               in_memory: bool = False, cpu_preload: bool = True, dq: bool = False, engine: Optional[str] = None, *,
               dtype: Optional[str] = None, weights: Optional[str] = None, seed: int = 1234,
               alignment_heads: Optional[Sequence[Tuple[int, int]]] = None, **model_kwargs) -> Whisper:
    """Same signature as stable_whisper.load_model (original_whisper.py:953-1009) plus keyword-only extensions.

    name     an official model name, a path to an upstream ``.pt`` checkpoint ({dims, model_state_dict}), or a
             HuggingFace Whisper checkpoint directory (config.json + model.safetensors; its generation config's
             alignment heads are used)
    dtype    'f16' (default, what the reference uses on a GPU) or 'f32' (strict parity with the reference's CPU path)
    weights  'random' -> seeded random initialisation at the architecture `name` (no network / no checkpoint offline)
    """
    if dq:
        raise NotImplementedError("dq (CPU dynamic quantisation, quantization.py:35-55) has no GPU counterpart")
    if engine not in (None, "amd", "mi355x"):
        raise NotImplementedError(f"engine={engine!r}: this package is a single MI355X-native engine (no dual backend)")
    device = "cuda:0" if device is None else str(device)
    if device == "cuda":
        device = "cuda:0"
    if not device.startswith("cuda"):
        raise RuntimeError("stable_ts_amd runs on an MI355X only (device='cuda[:i]'); there is no CPU path")
    dtype = dtype or "f16"
    sd = None
    if os.path.isdir(name) and os.path.isfile(os.path.join(name, "config.json")):
        dims, sd, hf_heads = read_hf_checkpoint(name)           # a HuggingFace checkpoint directory
        if alignment_heads is None:
            alignment_heads = hf_heads
    elif os.path.isfile(name):
        dims, sd = _read_checkpoint(name)
    elif name in _DIMS:
        dims = dims_for(name)
        root = download_root or os.path.join(os.path.expanduser("~"), ".cache", "whisper")
        path = os.path.join(root, name + ".pt")
        if weights != "random" and os.path.isfile(path):
            dims, sd = _read_checkpoint(path)
        elif weights != "random":
            raise RuntimeError(f"no checkpoint for {name!r} under {root} and no network to download one; "
                               f"pass a path, or weights='random' for seeded random weights")
    else:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    if alignment_heads is None and sd is not None and name in OFFICIAL_ALIGNMENT_HEADS:
        heads = OFFICIAL_ALIGNMENT_HEADS[name]                  # upstream load_model: model.set_alignment_heads(...)
        if all(l < dims.n_text_layer and h < dims.n_text_head for l, h in heads):
            alignment_heads = heads
    model = Whisper(dims, device=device, dtype=dtype, alignment_heads=alignment_heads, **model_kwargs)
    if sd is None:
        sd = random_state_dict(dims, seed=seed)
    model.load_state_dict(sd)
    return model
