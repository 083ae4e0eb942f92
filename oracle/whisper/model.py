"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of ``whisper.model`` (openai-whisper==20250625, the dependency
pinned by the reference: setup.py:30, whisper_compatibility.py:11-21).  Reference
call sites: decode.py:27-30,40 ; timing.py:50-61,105 ; alignment.py:405-429.

Pinning: encoder output / decoder logits / cross-attention qk are checked against
``transformers.WhisperModel`` loaded with the same weights through the name map
of the reference's ``whisper_word_level/hf_whisper.py:30-51`` in
``tests/test_oracle_pinning.py``.
"""
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn


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


class LayerNorm(nn.LayerNorm):
    def forward(self, x: Tensor) -> Tensor:
        return super().forward(x.float()).type(x.dtype)


class Linear(nn.Linear):
    def forward(self, x: Tensor) -> Tensor:
        return F.linear(x, self.weight.to(x.dtype), None if self.bias is None else self.bias.to(x.dtype))


class Conv1d(nn.Conv1d):
    def _conv_forward(self, x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
        return super()._conv_forward(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype))


def sinusoids(length, channels, max_timescale=10000):
    assert channels % 2 == 0
    log_timescale_increment = np.log(max_timescale) / (channels // 2 - 1)
    inv_timescales = torch.exp(-log_timescale_increment * torch.arange(channels // 2))
    scaled_time = torch.arange(length)[:, np.newaxis] * inv_timescales[np.newaxis, :]
    return torch.cat([torch.sin(scaled_time), torch.cos(scaled_time)], dim=1)


@contextmanager
def disable_sdpa():
    prev = MultiHeadAttention.use_sdpa
    try:
        MultiHeadAttention.use_sdpa = False
        yield
    finally:
        MultiHeadAttention.use_sdpa = prev


class MultiHeadAttention(nn.Module):
    use_sdpa = True

    def __init__(self, n_state: int, n_head: int):
        super().__init__()
        self.n_head = n_head
        self.query = Linear(n_state, n_state)
        self.key = Linear(n_state, n_state, bias=False)
        self.value = Linear(n_state, n_state)
        self.out = Linear(n_state, n_state)

    def forward(self, x: Tensor, xa: Optional[Tensor] = None, mask: Optional[Tensor] = None,
                kv_cache: Optional[dict] = None):
        q = self.query(x)
        if kv_cache is None or xa is None or self.key not in kv_cache:
            k = self.key(x if xa is None else xa)
            v = self.value(x if xa is None else xa)
        else:
            k = kv_cache[self.key]
            v = kv_cache[self.value]
        wv, qk = self.qkv_attention(q, k, v, mask)
        return self.out(wv), qk

    def qkv_attention(self, q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor] = None
                      ) -> Tuple[Tensor, Optional[Tensor]]:
        n_batch, n_ctx, n_state = q.shape
        scale = (n_state // self.n_head) ** -0.25
        q = q.view(*q.shape[:2], self.n_head, -1).permute(0, 2, 1, 3)
        k = k.view(*k.shape[:2], self.n_head, -1).permute(0, 2, 1, 3)
        v = v.view(*v.shape[:2], self.n_head, -1).permute(0, 2, 1, 3)

        if MultiHeadAttention.use_sdpa:
            a = F.scaled_dot_product_attention(q, k, v, is_causal=mask is not None and n_ctx > 1)
            out = a.permute(0, 2, 1, 3).flatten(start_dim=2)
            qk = None
        else:
            qk = (q * scale) @ (k * scale).transpose(-1, -2)
            if mask is not None:
                qk = qk + mask[:n_ctx, :n_ctx]
            qk = qk.float()
            w = F.softmax(qk, dim=-1).to(q.dtype)
            out = (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2)
            qk = qk.detach()
        return out, qk


class ResidualAttentionBlock(nn.Module):
    def __init__(self, n_state: int, n_head: int, cross_attention: bool = False):
        super().__init__()
        self.attn = MultiHeadAttention(n_state, n_head)
        self.attn_ln = LayerNorm(n_state)
        self.cross_attn = MultiHeadAttention(n_state, n_head) if cross_attention else None
        self.cross_attn_ln = LayerNorm(n_state) if cross_attention else None
        n_mlp = n_state * 4
        self.mlp = nn.Sequential(Linear(n_state, n_mlp), nn.GELU(), Linear(n_mlp, n_state))
        self.mlp_ln = LayerNorm(n_state)

    def forward(self, x: Tensor, xa: Optional[Tensor] = None, mask: Optional[Tensor] = None,
                kv_cache: Optional[dict] = None):
        x = x + self.attn(self.attn_ln(x), mask=mask, kv_cache=kv_cache)[0]
        if self.cross_attn:
            x = x + self.cross_attn(self.cross_attn_ln(x), xa, kv_cache=kv_cache)[0]
        x = x + self.mlp(self.mlp_ln(x))
        return x


class AudioEncoder(nn.Module):
    def __init__(self, n_mels: int, n_ctx: int, n_state: int, n_head: int, n_layer: int):
        super().__init__()
        self.conv1 = Conv1d(n_mels, n_state, kernel_size=3, padding=1)
        self.conv2 = Conv1d(n_state, n_state, kernel_size=3, stride=2, padding=1)
        self.register_buffer("positional_embedding", sinusoids(n_ctx, n_state))
        self.blocks: Iterable[ResidualAttentionBlock] = nn.ModuleList(
            [ResidualAttentionBlock(n_state, n_head) for _ in range(n_layer)])
        self.ln_post = LayerNorm(n_state)

    def forward(self, x: Tensor):
        x = F.gelu(self.conv1(x))
        x = F.gelu(self.conv2(x))
        x = x.permute(0, 2, 1)
        assert x.shape[1:] == self.positional_embedding.shape, "incorrect audio shape"
        x = (x + self.positional_embedding).to(x.dtype)
        for block in self.blocks:
            x = block(x)
        x = self.ln_post(x)
        return x


class TextDecoder(nn.Module):
    def __init__(self, n_vocab: int, n_ctx: int, n_state: int, n_head: int, n_layer: int):
        super().__init__()
        self.token_embedding = nn.Embedding(n_vocab, n_state)
        self.positional_embedding = nn.Parameter(torch.empty(n_ctx, n_state))
        self.blocks: Iterable[ResidualAttentionBlock] = nn.ModuleList(
            [ResidualAttentionBlock(n_state, n_head, cross_attention=True) for _ in range(n_layer)])
        self.ln = LayerNorm(n_state)
        mask = torch.empty(n_ctx, n_ctx).fill_(-np.inf).triu_(1)
        self.register_buffer("mask", mask, persistent=False)

    def forward(self, x: Tensor, xa: Tensor, kv_cache: Optional[dict] = None):
        offset = next(iter(kv_cache.values())).shape[1] if kv_cache else 0
        x = self.token_embedding(x) + self.positional_embedding[offset: offset + x.shape[-1]]
        x = x.to(xa.dtype)
        for block in self.blocks:
            x = block(x, xa, mask=self.mask, kv_cache=kv_cache)
        x = self.ln(x)
        logits = (x @ torch.transpose(self.token_embedding.weight.to(x.dtype), 0, 1)).float()
        return logits


class Whisper(nn.Module):
    def __init__(self, dims: ModelDimensions):
        super().__init__()
        self.dims = dims
        self.encoder = AudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state,
                                    dims.n_audio_head, dims.n_audio_layer)
        self.decoder = TextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state,
                                   dims.n_text_head, dims.n_text_layer)
        # default: all heads in the upper half of the decoder layers
        all_heads = torch.zeros(dims.n_text_layer, dims.n_text_head, dtype=torch.bool)
        all_heads[dims.n_text_layer // 2:] = True
        self.register_buffer("alignment_heads", all_heads.to_sparse(), persistent=False)

    def set_alignment_heads_mask(self, mask: Tensor):
        """mask: bool [n_text_layer, n_text_head] (upstream decodes it from a base85+gzip dump)."""
        assert mask.shape == (self.dims.n_text_layer, self.dims.n_text_head)
        self.register_buffer("alignment_heads", mask.bool().to_sparse(), persistent=False)

    def embed_audio(self, mel: Tensor):
        return self.encoder(mel)

    def logits(self, tokens: Tensor, audio_features: Tensor):
        return self.decoder(tokens, audio_features)

    def forward(self, mel: Tensor, tokens: Tensor):
        return self.decoder(tokens, self.encoder(mel))

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def is_multilingual(self):
        return self.dims.n_vocab >= 51865

    @property
    def num_languages(self):
        return self.dims.n_vocab - 51765 - int(self.is_multilingual)

    def install_kv_cache_hooks(self, cache: Optional[dict] = None):
        cache = {**cache} if cache is not None else {}
        hooks = []

        def save_to_cache(module, _, output):
            if module not in cache or output.shape[1] > self.dims.n_text_ctx:
                cache[module] = output
            else:
                cache[module] = torch.cat([cache[module], output], dim=1).detach()
            return cache[module]

        def install_hooks(layer: nn.Module):
            if isinstance(layer, MultiHeadAttention):
                hooks.append(layer.key.register_forward_hook(save_to_cache))
                hooks.append(layer.value.register_forward_hook(save_to_cache))

        self.decoder.apply(install_hooks)
        return cache, hooks

    def detect_language(self, mel: Tensor, tokenizer=None):
        from .decoding import detect_language as _detect
        return _detect(self, mel, tokenizer)


# ---- architecture table (upstream facts) ---------------------------------------
_DIMS = {
    #            mels  actx  astate ahead alayer vocab  tctx tstate thead tlayer
    "tiny.en":  (80,   1500, 384,   6,    4,     51864, 448, 384,   6,    4),
    "tiny":     (80,   1500, 384,   6,    4,     51865, 448, 384,   6,    4),
    "base.en":  (80,   1500, 512,   8,    6,     51864, 448, 512,   8,    6),
    "base":     (80,   1500, 512,   8,    6,     51865, 448, 512,   8,    6),
    "small.en": (80,   1500, 768,   12,   12,    51864, 448, 768,   12,   12),
    "small":    (80,   1500, 768,   12,   12,    51865, 448, 768,   12,   12),
    "medium.en": (80,  1500, 1024,  16,   24,    51864, 448, 1024,  16,   24),
    "medium":   (80,   1500, 1024,  16,   24,    51865, 448, 1024,  16,   24),
    "large-v2": (80,   1500, 1280,  20,   32,    51865, 448, 1280,  20,   32),
    "large-v3": (128,  1500, 1280,  20,   32,    51866, 448, 1280,  20,   32),
}


def dims_for(name: str) -> ModelDimensions:
    return ModelDimensions(*_DIMS[name])


def random_state_dict(dims: ModelDimensions, seed: int = 1234, std: float = 0.02,
                      embed_gain: float = 1.0, ts_gain: float = 1.0, ln_jitter: float = 0.0, xattn_gain: float = 1.0) -> Dict[str, Tensor]:
    """Deterministic random weights at the real architecture (no checkpoints exist offline).

    Linear / conv / embedding ~ N(0, std), biases ~ N(0, std), LN gamma=1 beta=0,
    decoder positions ~ N(0, 0.01).  The SAME routine feeds the oracle and the HIP path.
    """
    g = torch.Generator().manual_seed(seed)
    ref = Whisper(dims)
    sd = {}
    for k, v in ref.state_dict().items():
        if k.endswith("positional_embedding") and k.startswith("encoder"):
            sd[k] = v.clone()
        elif "_ln" in k or k.endswith("ln.weight") or k.endswith("ln.bias") or "ln_post" in k:
            sd[k] = torch.ones_like(v) if k.endswith("weight") else torch.zeros_like(v)
        elif k == "decoder.positional_embedding":
            sd[k] = torch.randn(v.shape, generator=g) * 0.01
        elif k == "decoder.token_embedding.weight":
            sd[k] = torch.randn(v.shape, generator=g) * std * embed_gain
            if ts_gain != 1.0:   # shrink the 1501 timestamp rows so that text tokens win more often (richer transcripts)
                sd[k][dims.n_vocab - 1501:] *= ts_gain
        else:
            sd[k] = torch.randn(v.shape, generator=g) * std
    if ln_jitter:
        # non-trivial LayerNorm affine parameters (gamma = 1 + j*N(0,1), beta = j*N(0,1)) from a generator of their own, so
        # that every other tensor is the same as without the jitter
        g2 = torch.Generator().manual_seed(seed + 7919)
        for k in sd:
            if "_ln" in k or k.endswith("ln.weight") or k.endswith("ln.bias") or "ln_post" in k:
                sd[k] = sd[k] + ln_jitter * torch.randn(sd[k].shape, generator=g2)
    if xattn_gain != 1.0:   # sharper cross-attention (scores scaled by xattn_gain): word timing on peaky attention maps
        for k in sd:
            if ".cross_attn.query.weight" in k or ".cross_attn.key.weight" in k:
                sd[k] = sd[k] * float(xattn_gain) ** 0.5
    return sd


def build_model(name_or_dims, seed: int = 1234, std: float = 0.02, embed_gain: float = 1.0, ts_gain: float = 1.0,
                ln_jitter: float = 0.0, xattn_gain: float = 1.0) -> Whisper:
    dims = dims_for(name_or_dims) if isinstance(name_or_dims, str) else name_or_dims
    model = Whisper(dims)
    model.load_state_dict(random_state_dict(dims, seed, std, embed_gain, ts_gain, ln_jitter, xattn_gain))
    return model.eval()
