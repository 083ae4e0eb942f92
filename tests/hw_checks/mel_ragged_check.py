"""Hardware check of swx_log_mel_ragged (csrc/swx_mel.hip, RAGGED kernels) against the oracle's
``pad_or_trim(log_mel_spectrogram(segment, padding=p), 3000)`` -- the spectrogram upstream computes for refine
(alignment.py:660-661, p = 0, floor from the batch max) and locate (alignment.py:924-925, p = 201).  Tolerance 1e-3 as
for swx_log_mel (the oracle's f32 FFT carries ~1e-4 of its own near the clamp floor); the frames past (n + p) // 160 must be exactly 0.0.  Exit code 0 = all cases agree.

    python tests/hw_checks/mel_ragged_check.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def synth(n, seed):
    g = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * (180 + 40 * seed) * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * g.standard_normal(n)
    return torch.from_numpy(x.astype(np.float32))


def main() -> int:
    import stable_ts_amd as sw
    from oracle.whisper.audio import N_FRAMES, log_mel_spectrogram, pad_or_trim
    bad = 0
    import dataclasses
    for n_mels in (80, 128):
        dims = dataclasses.replace(sw.dims_for("tiny.en"), n_mels=n_mels)    # only the filterbank matters here
        model = sw.Whisper(dims, dtype="f32", max_windows=3, max_rows=5)
        for n, padding in [(480000, 201), (480000, 0), (479999, 201), (163217, 201), (163217, 0), (8000, 0), (1601, 201),
                           (320, 0), (201, 0), (479840, 201)]:
            seg = synth(n, n % 7)
            want = pad_or_trim(log_mel_spectrogram(seg, dims.n_mels, padding=padding), N_FRAMES)
            got = model.log_mel_segments([seg], padding=padding)[0].cpu()
            k = (n + padding) // 160
            err = (got - want).abs().max().item()
            fill = bool((got[:, k:] == 0).all())
            print(f"n_mels={dims.n_mels} n={n} padding={padding}: max err {err:.2e}, fill exact {fill}")
            bad += (err > 1e-3) or not fill
        # click in the tail of a full chunk: the clamp floor comes from frame 3000, which pad_or_trim cuts
        seg = 1e-4 * synth(480000, 3)
        seg[-60:] = 0.9
        want = pad_or_trim(log_mel_spectrogram(seg, dims.n_mels, padding=201), N_FRAMES)
        err = (model.log_mel_segments([seg], padding=201)[0].cpu() - want).abs().max().item()
        print(f"n_mels={dims.n_mels} floor from the cut frame: max err {err:.2e}")
        bad += err > 1e-3
        # refine's batched call: ragged lengths are equal there, the floor is the batch max
        a, b = synth(52000, 1), 0.01 * synth(52000, 2)
        want = pad_or_trim(log_mel_spectrogram(torch.stack([a, b]), dims.n_mels), N_FRAMES)
        err = (model.log_mel_segments([a, b], batch_max=True).cpu() - want).abs().max().item()
        print(f"n_mels={dims.n_mels} batch max: max err {err:.2e}")
        bad += err > 1e-3
        # mixed lengths in one launch, per-item floor
        segs = [synth(480000, 5), synth(30001, 6), synth(777, 2)]
        got = model.log_mel_segments(segs, padding=201).cpu()
        for i, sg in enumerate(segs):
            want = pad_or_trim(log_mel_spectrogram(sg, dims.n_mels, padding=201), N_FRAMES)
            err = (got[i] - want).abs().max().item()
            print(f"n_mels={dims.n_mels} mixed batch item {i}: max err {err:.2e}")
            bad += err > 1e-3
        try:
            model.log_mel_segments([torch.zeros(200)])
            print("length 200 accepted (torch.stft rejects it)")
            bad += 1
        except RuntimeError:
            pass
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
