"""TEST INFRASTRUCTURE ONLY: a CPU stand-in for stable_ts_amd.engine.Engine / model.Whisper built on the oracle, so that
the HOST side of transcribe() (window state machine, temperature ladder, prompt handling, segment slicing, word
bookkeeping, silence suppression, regrouping) can be compared with the reference's own transcribe() on CPU -- the device
arithmetic is the same oracle on both sides, so any difference is a host-logic difference.  Never imported by the product.
"""
import types
import numpy as np
import torch

from oracle import stable as ost
from oracle.whisper.audio import N_FRAMES, N_SAMPLES, log_mel_spectrogram, pad_or_trim
from oracle.whisper.decoding import DecodingOptions
from oracle.whisper.timing import dtw as oracle_dtw
from oracle.whisper.tokenizer import get_tokenizer


import threading

_SDPA_LOCK = threading.Lock()      # the oracle's disable_sdpa() flips a class-level flag: serialise it across lanes (tests only)


class XKV:
    """stands in for the device cross-KV buffer: keeps the encoder output of the batch"""

    def __init__(self, xa: torch.Tensor):
        self.xa = xa
        self.n_windows = int(xa.shape[0])

    def select(self, idx):
        return XKV(self.xa[list(idx)])


class OracleEngine:
    tdtype = torch.float32
    dtype_name = "f32"
    device = torch.device("cpu")

    def __init__(self, oracle_model):
        self.m = oracle_model
        self.dims = oracle_model.dims
        self.tok = get_tokenizer(oracle_model.is_multilingual, num_languages=oracle_model.num_languages, language="en",
                                 task="transcribe")
        self.n_decode_calls = 0

    def cross_kv(self, xa):
        return XKV(xa)

    @torch.no_grad()
    def decode(self, xkv, init_tokens, *, n_group=1, beam=False, temperature=0.0, patience=None, sample_len=224,
               sot_index=0, suppress_blank=True, apply_timestamp_rules=True, max_initial_timestamp_index=None, eot=0, sot=0,
               no_timestamps=-1, timestamp_begin=0, no_speech=-1, blank_token=-1, suppress_tokens=(), ts_mask=None,
               min_tokens=0, seed=0, window_uid=None, torch_rng=False):
        # (torch_rng: this stand-in always samples through upstream's Categorical, i.e. from torch's generator)
        self.n_decode_calls += 1
        W = xkv.n_windows
        TS = self.dims.n_text_ctx + 1
        toks = np.zeros((W, 1, TS), dtype=np.int32)
        lens = np.zeros((W, 1), dtype=np.int32)
        slp = np.zeros((W, 1), dtype=np.float32)
        nsp = np.zeros(W, dtype=np.float32)
        opts = DecodingOptions(
            fp16=False, language="en", temperature=temperature, sample_len=sample_len,
            beam_size=n_group if beam else None, best_of=(n_group if (not beam and n_group > 1) else None),
            patience=patience, suppress_blank=suppress_blank, suppress_tokens=list(suppress_tokens),
            without_timestamps=not apply_timestamp_rules,
            max_initial_timestamp=(None if max_initial_timestamp_index is None else max_initial_timestamp_index * 0.02))
        for w in range(W):
            init = tuple(int(t) for t in init_tokens[w])

            class _Task(ost.DecodingTaskStable):
                def _get_initial_tokens(self_inner):
                    return init

            mask = None if ts_mask is None else ts_mask[w].bool()
            task = _Task(self.m, opts, ts_token_mask=mask, audio_features=xkv.xa[w:w + 1])
            assert task.sot_index == sot_index and task.sample_begin == len(init)
            if min_tokens:
                pos = len(task.logit_filters) - (0 if opts.without_timestamps else 1)
                task.logit_filters.insert(pos, ost._MinTokens(task.tokenizer.eot, task.sample_begin, min_tokens))
            with _SDPA_LOCK:               # keep another lane's scoring pass from flipping the attention code path mid-decode
                r = task.run(torch.zeros(1, self.dims.n_mels, N_FRAMES))[0]
            n = len(r.tokens)
            toks[w, 0, len(init): len(init) + n] = r.tokens
            lens[w, 0] = n
            slp[w, 0] = r.avg_logprob * (n + 1)
            nsp[w] = r.no_speech_prob
        return dict(tokens=toks, lens=lens, sum_logprobs=slp, no_speech_prob=nsp, steps=sample_len, sample_begin=len(init_tokens[0]))

    @torch.no_grad()
    def score(self, xkv, tokens, n_frames, n_sot, eot, qk_scale=1.0, medfilt_width=7):
        W = len(tokens)
        max_n = max(len(t) for t in tokens)
        neg = torch.zeros(W, max_n, self.dims.n_audio_ctx)
        probs, T = [], []
        from oracle.whisper.model import disable_sdpa
        from oracle.whisper.timing import median_filter
        for w, tk in enumerate(tokens):
            text = list(tk[n_sot + 1:-1])
            qks = [None] * self.dims.n_text_layer
            hooks = [blk.cross_attn.register_forward_hook(lambda _, i, o, k=k: qks.__setitem__(k, o[-1]))
                     for k, blk in enumerate(self.m.decoder.blocks)]
            with _SDPA_LOCK, disable_sdpa():
                logits = self.m.decoder(torch.tensor([list(tk)]), xkv.xa[w:w + 1])[0]
            for h in hooks:
                h.remove()
            p = logits[n_sot:, :eot].softmax(dim=-1)                 # timing.py:62-64; targets >= eot score 0 like the kernel
            probs.append([float(p[i, t]) if t < eot else 0.0 for i, t in enumerate(text)])
            wts = torch.cat([qks[l][:, h] for l, h in self.m.alignment_heads.indices().T], dim=0)
            wts = wts[:, n_sot:-1, :n_frames[w]]
            wts = (wts * qk_scale).softmax(dim=-1)
            std, mean = torch.std_mean(wts, dim=-2, keepdim=True, unbiased=False)
            mat = -median_filter((wts - mean) / std, medfilt_width).mean(dim=0)     # [rows, n_frames]
            neg[w, :mat.shape[0], :mat.shape[1]] = mat
            T.append(len(text))
        return probs, neg, T

    @torch.no_grad()
    def score_start(self, *args, **kw):
        """the product enqueues the scoring pass before it splits the words and collects it afterwards; the stand-in has no
        queue: it runs the pass here and hands the result over in `score_finish`"""
        return self.score(*args, **kw)

    @staticmethod
    def score_finish(handle):
        return handle

    def score_qk(self, xkv, tokens, *, n_sot, eot, row0, n_rows):
        """raw scores of the alignment heads (or of every head on the ``all_heads()`` view), rows row0 .. row0+n_rows-1"""
        from oracle.whisper.model import disable_sdpa
        W = len(tokens)
        heads = ([(l, h) for l in range(self.dims.n_text_layer) for h in range(self.dims.n_text_head)] if self._every_head
                 else [tuple(int(v) for v in p) for p in self.m.alignment_heads.indices().T])
        out = torch.zeros(W, len(heads), n_rows, self.dims.n_audio_ctx)
        probs = []
        for w, tk in enumerate(tokens):
            text = list(tk[n_sot + 1:-1])
            qks = [None] * self.dims.n_text_layer
            hooks = [blk.cross_attn.register_forward_hook(lambda _, i, o, k=k: qks.__setitem__(k, o[-1]))
                     for k, blk in enumerate(self.m.decoder.blocks)]
            with _SDPA_LOCK, disable_sdpa():
                logits = self.m.decoder(torch.tensor([list(tk)]), xkv.xa[w:w + 1])[0]
            for h in hooks:
                h.remove()
            p = logits[n_sot:, :eot].softmax(dim=-1)
            probs.append([float(p[i, t]) if t < eot else 0.0 for i, t in enumerate(text)])
            rows = min(n_rows, len(tk) - row0)
            for k, (l, h) in enumerate(heads):
                out[w, k, :rows] = qks[l][0, h, row0: row0 + rows]
        return probs, out

    _every_head = False

    def all_heads(self):
        view = OracleEngine(self.m)
        view._every_head = True
        return view

    @property
    def n_alignment_heads(self):
        return int(self.m.alignment_heads.indices().shape[1])

    # ---- the head-selection variants (product: csrc/swx_headsel.hip through Engine.score_q / heads_dynamic / heads_new /
    #      pool_matrices).  The stand-in computes them the way the REFERENCE writes them (timing.py:87-163) from every head's
    #      scores, so the CPU tests compare the product's control flow with the reference's on identical arithmetic.
    def score_q(self, xkv, tokens, *, n_sot, eot):
        n = len(tokens)
        probs, qk = self.all_heads().score_qk(xkv, [list(tokens)], n_sot=n_sot, eot=eot, row0=0, n_rows=n)
        return dict(qk=qk[0], n=n, n_sot=n_sot, probs=probs[0])

    def heads_dynamic(self, st, n_frames, *, count, qk_scale=1.0, medfilt_width=7, jump_indices=None):
        n, n_sot, F = st["n"], st["n_sot"], int(n_frames)
        qk = (st["qk"][:, n_sot:n - 1, :F] * qk_scale).softmax(dim=-1)                     # [heads, T+1, F]
        if jump_indices is None:
            peaks = qk.topk(1, dim=-1).indices
        else:
            j = np.pad(np.asarray(jump_indices), (0, 1), constant_values=F)
            peaks = torch.from_numpy(j[:-1] + ((j[1:] - j[:-1]) * 0.5))[None, :, None]
        distances = (peaks.expand_as(qk) - torch.arange(qk.size(-1))).abs() / 1500
        scores = (distances * qk).sum(dim=-1)
        heads = [sc.topk(count, largest=False).indices for sc in scores.T]
        weights = torch.stack([qk[h, i] for i, h in enumerate(heads)], dim=1)
        std, mean = torch.std_mean(weights, dim=-2, keepdim=True, unbiased=False)
        neg = torch.zeros(weights.shape[1], self.dims.n_audio_ctx)
        neg[:, :F] = -self.median_filter((weights - mean) / std, medfilt_width).mean(dim=0)
        return neg

    def heads_new(self, st, n_frames, *, qk_scale=1.0, medfilt_width=7, topk=20, w_colnorm=1, w_rownorm=1, w_coverage=0):
        n, n_sot, F = st["n"], st["n_sot"], int(n_frames)
        L, H = self.dims.n_text_layer, self.dims.n_text_head
        w = st["qk"].reshape(L, H, n, -1)[..., :F]
        w = (self.median_filter(w, medfilt_width) * qk_scale).softmax(dim=-1)
        score = torch.zeros(L, H)
        if w_colnorm > 0:
            score += w_colnorm * w.norm(dim=-2).sum(-1)
        if w_rownorm > 0:
            score += w_rownorm * w.norm(dim=-1).sum(-1)
        if w_coverage > 0:
            coverage = torch.sum(w, dim=2)
            penalty = torch.max(coverage, coverage.clone().fill_(0.5)).sum(-1) - coverage.size(-1) * 0.5
            score -= w_coverage * penalty
        top = score.flatten().topk(topk).indices
        m = w[top // H, top % H]
        m = torch.mean(m / m.norm(dim=-2, keepdim=True), 0)
        neg = torch.zeros(n - 1 - n_sot, self.dims.n_audio_ctx)
        neg[:, :F] = -m[n_sot:-1]
        return neg

    def pool_matrices(self, negs, n_heads):
        tot = float(sum(n_heads))
        out = torch.zeros_like(negs[0])
        for x, h in zip(negs, n_heads):
            out += (h / tot) * x
        return out

    def median_filter(self, x, width):
        from oracle.whisper.timing import median_filter
        return median_filter(x, width)

    @torch.no_grad()
    def forward_logits(self, xkv, tokens, pad_token=0):
        n = max(len(t) for t in tokens)
        out = torch.zeros(len(tokens), n, self.dims.n_vocab)
        for w, t in enumerate(tokens):
            out[w, :len(t)] = self.m.decoder(torch.tensor([list(t)]), xkv.xa[w:w + 1])[0]
        return out

    def dtw(self, neg, N, M):
        out = []
        for w in range(neg.shape[0]):
            ti, tj = oracle_dtw(neg[w, :N[w], :M[w]])
            out.append((np.asarray(ti), np.asarray(tj)))
        return out


class CpuWhisper:
    """the attributes and methods of stable_ts_amd.model.Whisper that the host code reads"""

    computes_on_host = True      # the product parks torch's intra-op pool inside its entry points; here the pool is the compute

    def __init__(self, oracle_model):
        from stable_ts_amd.transcribe import transcribe_stable
        self.om = oracle_model
        self.dims = oracle_model.dims
        self.is_multilingual = oracle_model.is_multilingual
        self.num_languages = oracle_model.num_languages
        self.device = torch.device("cpu")
        self.engine = OracleEngine(oracle_model)
        self.transcribe = types.MethodType(transcribe_stable, self)
        from stable_ts_amd.transcribe import transcribe_minimal
        self.transcribe_minimal = types.MethodType(transcribe_minimal, self)

    def log_mel_batch(self, audios, paddings=None):
        out = []
        for b, a in enumerate(audios):
            pad = 0 if paddings is None else paddings[b]
            assert a.shape[-1] + pad == N_SAMPLES
            out.append(pad_or_trim(log_mel_spectrogram(torch.as_tensor(a, dtype=torch.float32), self.dims.n_mels, padding=pad), N_FRAMES))
        return torch.stack(out)

    def log_mel(self, audio, padding=0):
        return self.log_mel_batch([audio], [padding])[0]

    def log_mel_segments(self, audios, padding=0, batch_max=False):
        audios = [torch.as_tensor(a, dtype=torch.float32) for a in audios]
        if batch_max:                                      # upstream's batched call: one clamp floor for the batch
            return pad_or_trim(log_mel_spectrogram(torch.stack(audios), self.dims.n_mels, padding=padding), N_FRAMES)
        return torch.stack([pad_or_trim(log_mel_spectrogram(a, self.dims.n_mels, padding=padding), N_FRAMES) for a in audios])

    # the reference encodes inside ``disable_sdpa()`` in its alignment flows (timing.py:58-60) and with SDPA in transcribe
    # (decode.py:27-30); the two attention code paths of the oracle differ by ~1e-7, enough to move a DTW path on the
    # near-uniform attention of random weights.  Tests of align / align_words / refine / locate set this to True.
    manual_attention_encoder = False

    @torch.no_grad()
    def encoder(self, mel):
        with _SDPA_LOCK:
            if self.manual_attention_encoder:
                from oracle.whisper.model import disable_sdpa
                with disable_sdpa():
                    return self.om.encoder(mel)
            return self.om.encoder(mel)

    def cross_kv(self, xa):
        return XKV(xa)

    @torch.no_grad()
    def detect_language(self, mel):
        from oracle.whisper.decoding import detect_language
        return detect_language(self.om, mel)

    def clone_for_stream(self):
        import copy
        return CpuWhisper(copy.deepcopy(self.om))      # the oracle's hook-based KV cache is per module instance

    def stream_context(self):
        import contextlib
        return contextlib.nullcontext(), None


def install(monkeypatch):
    """route the product's device-buffer helper to the stand-in's"""
    import stable_ts_amd.transcribe as T
    monkeypatch.setattr(T, "_xkv_select", lambda model, xkv, idx: xkv.select(idx))
