#!/usr/bin/env python
"""Where the fp16 mode's per-token log-probability error comes from: the f32 CPU oracle of the full-depth
large-v3 architecture (bench weights) with fp16 ROUNDING emulated at one class of places at a time -- stored weights, the
residual stream, the activations a GEMM reads / writes, the final LayerNorm's output -- in the encoder, the decoder or both.
Everything else stays f32, so each line isolates one contribution; "all" is the emulation of the GPU's fp16 mode (fp16 storage,
f32 accumulation / LayerNorm / softmax).  Test infrastructure: runs on the CPU, imports oracle/.
usage: python scripts/f16_error_budget.py [--tokens 24] [--out profiles/r04_f16_error_budget.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stable as ost                      # noqa: E402
from oracle.whisper import model as om                # noqa: E402
from oracle.whisper.decoding import DecodingOptions   # noqa: E402

R = lambda t: t.half().float()
FLAGS = dict(enc=set(), dec=set())


def _patch():
    def block_forward(self, x, xa=None, mask=None, kv_cache=None):
        side = FLAGS["dec" if self.cross_attn else "enc"]
        rr = R if "res" in side else (lambda t: t)
        x = rr(x + self.attn(self.attn_ln(x), mask=mask, kv_cache=kv_cache)[0])
        if self.cross_attn:
            x = rr(x + self.cross_attn(self.cross_attn_ln(x), xa, kv_cache=kv_cache)[0])
        x = rr(x + self.mlp(self.mlp_ln(x)))
        return x
    om.ResidualAttentionBlock.forward = block_forward


def _hooks(model):
    hs = []

    def mk(side):
        def pre(mod, args):
            return (R(args[0]),) + tuple(args[1:]) if "act" in FLAGS[side] else None

        def post(mod, args, out):
            return R(out) if "act" in FLAGS[side] else None
        return pre, post
    for side, root in (("enc", model.encoder), ("dec", model.decoder)):
        pre, post = mk(side)
        for m in root.modules():
            if isinstance(m, (om.Linear, om.Conv1d)):
                hs.append(m.register_forward_pre_hook(pre))
                hs.append(m.register_forward_hook(post))
    # final LayerNorm of the decoder: its output is the A operand of the vocabulary GEMM
    hs.append(model.decoder.ln.register_forward_hook(lambda mod, a, out: R(out) if "ln_out" in FLAGS["dec"] else None))
    return hs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=24)
    ap.add_argument("--out", default="")
    ap.add_argument("--words", action="store_true", help="also: the word-timestamp stage (timing.py:202-306) on the decoded text "
                    "under each variant -- DTW row starts / word times against the f32 ones")
    ap.add_argument("--only", default="", help="comma-separated substrings: run only the variants whose name contains one")
    ap.add_argument("--recipe", default="", help="overrides of BENCH_WEIGHTS, e.g. xattn_gain=16,ts_gain=0.1")
    ap.add_argument("--beam", type=int, default=0)
    ap.add_argument("--decode-check", action="store_true", help="also DECODE under the emulated fp16 mode and compare the tokens")
    args = ap.parse_args()
    torch.set_num_threads(8)
    import stable_ts_amd.model as pm
    recipe = dict(getattr(pm, "BENCH_WEIGHTS", dict(embed_gain=9.0, ts_gain=0.01, ln_jitter=0.1, xattn_gain=8.0)))
    for kv in filter(None, args.recipe.split(",")):
        k_, v_ = kv.split("=")
        recipe[k_] = float(v_)
    print("recipe", recipe, flush=True)
    dims = om.dims_for("large-v3")
    sd = om.random_state_dict(dims, 1234, 0.02, **recipe)
    _patch()
    m = om.Whisper(dims)
    m.load_state_dict(sd)
    m.eval()
    _hooks(m)
    g = torch.Generator().manual_seed(7)
    t = torch.linspace(0, 1, 3000)
    mel = (torch.sin(t[None, :] * (5 + torch.arange(128)[:, None] * 0.37)) * 0.5 + 0.3 * torch.randn(128, 3000, generator=g)).float()
    t0 = time.time()
    with torch.no_grad():
        xa = m.encoder(mel[None])
    print(f"encoder f32: {time.time() - t0:.1f}s", flush=True)
    opts = DecodingOptions(fp16=False, language="en", max_initial_timestamp=None, sample_len=args.tokens,
                           **(dict(beam_size=args.beam) if args.beam > 1 else {}))
    ref, _ = ost.decode_stable(m, mel, opts, audio_features=xa, min_tokens=args.tokens)
    task = ost.DecodingTaskStable(m, opts)
    seq = list(task.initial_tokens) + list(ref.tokens)
    n0 = len(task.initial_tokens)
    toks = torch.tensor([seq])

    def logp(xa_):
        with torch.no_grad():
            lg = m.decoder(toks, xa_)[0].double()
        lp = torch.log_softmax(lg, -1)
        idx = torch.tensor(seq[n0:])
        return lp[n0 - 1: len(seq) - 1].gather(1, idx[:, None])[:, 0].numpy(), lg.numpy()

    base_lp, base_lg = logp(xa)
    w32 = {k: v.clone() for k, v in m.state_dict().items()}

    def set_weights(rounded_enc, rounded_dec):
        sd2 = {}
        for k, v in w32.items():
            is_mat = v.ndim >= 2 and "positional" not in k
            rd = (rounded_enc and k.startswith("encoder.")) or (rounded_dec and k.startswith("decoder."))
            sd2[k] = R(v) if (is_mat and rd) else v
        m.load_state_dict(sd2)

    variants = [
        ("weights fp16 (encoder + decoder)", dict(w=(1, 1))),
        ("residual stream fp16, decoder", dict(dec={"res"})),
        ("residual stream fp16, encoder", dict(enc={"res"})),
        ("GEMM inputs / outputs fp16, decoder", dict(dec={"act"})),
        ("GEMM inputs / outputs fp16, encoder", dict(enc={"act"})),
        ("final LayerNorm output fp16", dict(dec={"ln_out"})),
        ("all of the above = the fp16 mode", dict(w=(1, 1), enc={"res", "act"}, dec={"res", "act", "ln_out"})),
        ("fp16 mode with an f32 residual stream in the decoder", dict(w=(1, 1), enc={"res", "act"}, dec={"act", "ln_out"})),
        ("fp16 mode with f32 residual streams in encoder and decoder", dict(w=(1, 1), enc={"act"}, dec={"act", "ln_out"})),
        ("fp16 mode, f32 residual streams, f32 final LayerNorm output", dict(w=(1, 1), enc={"act"}, dec={"act"})),
    ]
    out = dict(tokens=len(ref.tokens), token_logprob_range=[float(base_lp.min()), float(base_lp.max())], variants={})
    xa_cache = {}
    from oracle.whisper.tokenizer import get_tokenizer
    tok = get_tokenizer(True, num_languages=m.num_languages, language="en", task="transcribe")
    mask = torch.zeros(dims.n_text_layer, dims.n_text_head, dtype=torch.bool)
    for l, h in ((7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6)):
        mask[l, h] = True
    m.set_alignment_heads_mask(mask)
    text = [t_ for t_ in ref.tokens if t_ < tok.eot]

    def words_of(xa_):
        w, cache = ost.find_alignment(m, tok, list(text), mel, 480000, audio_features=xa_, return_cache=True)
        i, j = cache["dtw_path"]
        first = {int(r): int(c) for r, c in reversed(list(zip(i.tolist(), j.tolist())))}
        return w, first
    base_words = words_of(xa) if args.words else None
    if args.decode_check:
        FLAGS["enc"], FLAGS["dec"] = {"res", "act"}, {"res", "act", "ln_out"}
        set_weights(1, 1)
        with torch.no_grad():
            xa16 = m.encoder(mel[None])
        got, _ = ost.decode_stable(m, mel, opts, audio_features=xa16, min_tokens=args.tokens)
        n_same = 0
        for a_, b_ in zip(got.tokens, ref.tokens):
            if a_ != b_:
                break
            n_same += 1
        out["decode_check"] = dict(beam=args.beam, tokens=len(ref.tokens), identical_prefix=n_same,
                                   first_tokens=(got.tokens[:3], ref.tokens[:3]),
                                   d_avg_logprob=abs(got.avg_logprob - ref.avg_logprob))
        print("decode under emulated fp16:", json.dumps(out["decode_check"]), flush=True)
        FLAGS["enc"], FLAGS["dec"] = set(), set()
        set_weights(0, 0)
    if args.only:
        variants = [(n_, v_) for n_, v_ in variants if any(o in n_ for o in args.only.split(","))]
    print("text tokens", len(text), "logit gap min", float(np.sort(base_lg[n0 - 1: len(seq) - 1], -1)[:, -1].min()), flush=True)
    for name, v in variants:
        FLAGS["enc"], FLAGS["dec"] = set(v.get("enc", ())), set(v.get("dec", ()))
        w = v.get("w", (0, 0))
        set_weights(*w)
        key = (w[0], tuple(sorted(FLAGS["enc"])))
        if key not in xa_cache:
            with torch.no_grad():
                xa_cache[key] = m.encoder(mel[None])
        lp, lg = logp(xa_cache[key])
        d = np.abs(lp - base_lp)
        unsat = base_lp > np.log(0.05)
        top2 = np.sort(base_lg[n0 - 1: len(seq) - 1], -1)[:, -2:]
        rec = dict(max_dlogp=float(d.max()), mean_dlogp=float(d.mean()),
                   max_dlogp_p_gt_0p05=float(d[unsat].max()) if unsat.any() else None,
                   d_avg_logprob=float(abs(lp.sum() - base_lp.sum()) / (len(lp) + 1)),
                   max_dlogit=float(np.abs(lg - base_lg)[n0 - 1: len(seq) - 1].max()),
                   argmax_same=bool((lg[n0 - 1: len(seq) - 1].argmax(-1) == base_lg[n0 - 1: len(seq) - 1].argmax(-1)).all()),
                   min_top1_top2_gap=float((top2[:, 1] - top2[:, 0]).min()))
        if args.words:
            w, first = words_of(xa_cache[key])
            bw, bfirst = base_words
            dt = np.asarray([(abs(a.start - b.start), abs(a.end - b.end)) for a, b in zip(w, bw)])
            rec.update(words=len(bw), words_within_20ms=float(((dt[:, 0] <= 0.0201) & (dt[:, 1] <= 0.0201)).mean()),
                       max_word_dt=float(dt.max()), dtw_row_start_max_frame_diff=int(max(abs(first[r] - bfirst[r]) for r in bfirst)),
                       dtw_rows_moved=int(sum(first[r] != bfirst[r] for r in bfirst)))
        out["variants"][name] = rec
        print(name, json.dumps(rec), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
