"""Hardware check of swx_score_qk (raw per-head attention scores of the scoring pass) and of the head-selection kernels
(csrc/swx_headsel.hip through Engine.score_q / heads_dynamic / heads_new / pool_matrices: dynamic heads, the 'new' aligner,
pooled models), strict f32 mode, against the CPU oracle: raw scores within 2e-4; the variants' NEGATED matrices within 2e-3 of
the reference's tensor expressions run on the oracle's every-head scores (tests/oracle_engine.py); DTW paths and word times of
the variants equal to the same host code run on that stand-in.  Exit code 0 = parity.

    python tests/hw_checks/score_qk_check.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.join(TESTS, "golden"))


def main() -> int:
    import stable_ts_amd as sw
    import stable_ts_amd.transcribe as T
    from make_golden import synth_audio
    from oracle.whisper import model as om
    from oracle_engine import CpuWhisper
    from stable_ts_amd.timing import AlignmentJob, find_alignment_batch
    from stable_ts_amd.tokenizer import get_tokenizer
    dims = sw.dims_for("tiny.en")
    sd = sw.random_state_dict(dims, seed=1234, std=0.02, embed_gain=2.0, ts_gain=0.5)
    model = sw.Whisper(dims, device="cuda:0", dtype="f32", max_windows=2, max_rows=5)
    model.load_state_dict(sd)
    ref = om.Whisper(om.ModelDimensions(**dims.__dict__))
    ref.load_state_dict(sd)
    ref.eval()
    mask = torch.zeros(dims.n_text_layer, dims.n_text_head, dtype=torch.bool)
    for l, h in model.engine.alignment_heads:
        mask[l, h] = True
    ref.set_alignment_heads_mask(mask)
    cpu = CpuWhisper(ref)
    tok = get_tokenizer(False, num_languages=model.num_languages)
    audio = torch.as_tensor(synth_audio(30.0, seed=2))
    mel = model.log_mel(audio)
    xkv = model.cross_kv(model.encoder(mel[None]))
    xkv_cpu = cpu.cross_kv(cpu.encoder(mel.cpu()[None]))
    text = tok.encode(" aaat aaau aaax. aabc aaat, aaau aaax aabc aaat.")
    bad = 0
    n_sot = len(tok.sot_sequence)
    ids = [*tok.sot_sequence, tok.no_timestamps, *text, tok.eot]
    for name, dev_eng, cpu_eng in (("alignment heads", model.engine, cpu.engine),):
        for row0, rows in ((n_sot, len(ids) - n_sot - 1), (0, len(ids))):
            p_d, qk_d = dev_eng.score_qk(xkv, [ids], n_sot=n_sot, eot=tok.eot, row0=row0, n_rows=rows)
            p_c, qk_c = cpu_eng.score_qk(xkv_cpu, [ids], n_sot=n_sot, eot=tok.eot, row0=row0, n_rows=rows)
            err = (qk_d.cpu() - qk_c).abs().max().item()
            perr = float(np.abs(np.asarray(p_d[0]) - np.asarray(p_c[0])).max())
            print(f"{name}, rows {row0}..{row0 + rows - 1}: qk {tuple(qk_d.shape)} max err {err:.2e}, token prob err {perr:.2e}")
            bad += (err > 2e-4) or (perr > 1e-4) or tuple(qk_d.shape) != tuple(qk_c.shape)
    # the kernels' matrices against the reference's expressions on the oracle's scores of EVERY head
    F = 1500
    st_d = model.engine.score_q(xkv, ids, n_sot=n_sot, eot=tok.eot)
    st_c = cpu.engine.score_q(xkv_cpu, ids, n_sot=n_sot, eot=tok.eot)
    perr = float(np.abs(np.asarray(st_d["probs"]) - np.asarray(st_c["probs"])).max())
    print(f"score_q: token prob err {perr:.2e}")
    bad += perr > 1e-4
    jump = np.linspace(10, 1400, len(ids) - n_sot - 1).round().astype(np.int64)
    for label, call in (("heads_dynamic(count=4)", lambda e, st: e.heads_dynamic(st, F, count=4)),
                        ("heads_dynamic(count=3, jump midpoints)", lambda e, st: e.heads_dynamic(st, F, count=3, jump_indices=jump)),
                        ("heads_dynamic(count=5, 1234 frames, qk_scale 2)", lambda e, st: e.heads_dynamic(st, 1234, count=5, qk_scale=2.0)),
                        ("heads_new()", lambda e, st: e.heads_new(st, F)),
                        ("heads_new(topk=7, coverage, width 5, 987 frames)", lambda e, st: e.heads_new(st, 987, topk=7, w_coverage=0.5, medfilt_width=5))):
        nd, nc = call(model.engine, st_d).cpu(), call(cpu.engine, st_c)
        err = (nd - nc).abs().max().item()
        print(f"{label}: matrix {tuple(nd.shape)} max err {err:.2e}")
        bad += (err > 2e-3) or tuple(nd.shape) != tuple(nc.shape)
    a, b = torch.randn(40, 1500), torch.randn(40, 1500)
    pd = model.engine.pool_matrices([a.cuda(), b.cuda()], [6, 4]).cpu()
    bad += (pd - (0.6 * a + 0.4 * b)).abs().max().item() > 1e-6
    T_select = T._xkv_select
    for kw in (dict(dynamic_heads=4), dict(dynamic_heads="3,2"), dict(aligner="new")):
        job_d, job_c = AlignmentJob(tok, list(text), 480000), AlignmentJob(tok, list(text), 480000)
        words_d = find_alignment_batch(model, [job_d], xkv, return_debug=True, **kw)[0]
        T._xkv_select = lambda m, x, idx: x.select(idx)
        try:
            words_c = find_alignment_batch(cpu, [job_c], xkv_cpu, return_debug=True, **kw)[0]
        finally:
            T._xkv_select = T_select
        same_path = all(np.array_equal(a, b) for a, b in zip(job_d.debug["path"], job_c.debug["path"]))
        dt = max(max(abs(a.start - b.start), abs(a.end - b.end)) for a, b in zip(words_d, words_c))
        print(f"{kw}: dtw path equal {same_path}, worst word-time difference {dt:.3f} s")
        bad += (not same_path) or dt > 0.02
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
